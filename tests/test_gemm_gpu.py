"""HIP MFMA GEMM (csrc/gemm.hip) vs a plain PyTorch fp32 reference, through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, b, a_km, b_km):
    A = a.float().t() if a_km else a.float()
    B = b.float().t() if b_km else b.float()
    return A.double() @ B.double().t()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("a_km,b_km", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 96, 96), (393, 288, 192), (1000, 136, 64), (24, 24, 512)])
def test_gemm_layouts(dtype, a_km, b_km, M, N, K):
    from deepcoro_clip_amd import kernels as Kx

    if dtype == torch.bfloat16 and ((a_km and M % 8) or (b_km and N % 8) or N % 8):
        pytest.skip("bf16 callers always meet the 16-byte chunk rules; the unaligned variant exists for fp32 only")
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if a_km else (M, K), generator=g).to("cuda", dtype)
    b = torch.randn((K, N) if b_km else (N, K), generator=g).to("cuda", dtype)
    out = Kx.gemm(a, b, a_km=a_km, b_km=b_km, out_dtype=torch.float32)
    ref = _ref(a, b, a_km, b_km)
    # integer-exact check of the fragment maps first (small ints are exact in bf16 and fp32)
    ai = torch.randint(-3, 4, a.shape, generator=g).to("cuda", dtype)
    bi = torch.randint(-3, 4, b.shape, generator=g).to("cuda", dtype)
    outi = Kx.gemm(ai, bi, a_km=a_km, b_km=b_km, out_dtype=torch.float32)
    assert torch.equal(outi.double(), _ref(ai, bi, a_km, b_km)), "fragment layout mismatch (exact integer data)"
    tol = 1e-4 if dtype == torch.float32 else 1e-5  # bf16 inputs are exact; accumulation is fp32 in both
    err = (out.double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err < tol, err


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dtype):
    from deepcoro_clip_amd import kernels as Kx

    g = torch.Generator(device="cpu").manual_seed(5)
    M, N, K = 300, 192, 96
    a = torch.randn(M, K, generator=g).to("cuda", dtype)
    w = (torch.randn(N, K, generator=g) * 0.1).to("cuda", dtype)
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to("cuda", dtype)
    pre = torch.empty(M, N, device="cuda", dtype=dtype)
    out = Kx.gemm(a, w, bias=bias, gelu=True, save_pre=True, aux=pre, resid=res)
    ref_pre = a.float() @ w.float().t() + bias
    ref = torch.nn.functional.gelu(ref_pre) + res.float()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (pre.float() - ref_pre).abs().max() < tol * max(1.0, ref_pre.abs().max().item())
    assert (out.float() - ref).abs().max() < tol * max(1.0, ref.abs().max().item())
    # dgelu epilogue: out = (a @ w^T) * gelu'(pre)
    out2 = Kx.gemm(a, w, dgelu=True, aux=pre)
    x = pre.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    ref2 = (a.float() @ w.float().t()) * x.grad
    assert (out2.float() - ref2).abs().max() < tol * max(1.0, ref2.abs().max().item())


def test_gemm_fused_bias_gradient():
    from deepcoro_clip_amd import kernels as Kx

    g = torch.Generator(device="cpu").manual_seed(11)
    rows, N, K = 3000, 192, 96
    dy = torch.randn(rows, N, generator=g).to("cuda", torch.bfloat16)
    x = torch.randn(rows, K, generator=g).to("cuda", torch.bfloat16)
    db = torch.zeros(N, device="cuda")
    dw = Kx.gemm(dy, x, a_km=True, b_km=True, out_dtype=torch.float32, splitk=5, colsum=db)
    assert (db.double() - dy.double().sum(0)).abs().max() < 1e-2
    assert (dw.double() - dy.double().t() @ x.double()).abs().max() < 1e-2


def test_gemm_small_tiles_and_gelu_precision():
    from deepcoro_clip_amd import kernels as Kx

    g = torch.Generator(device="cpu").manual_seed(12)
    a = torch.randn(768, 768, generator=g).cuda()
    w = (torch.randn(768, 768, generator=g) * 0.05).cuda()
    out = Kx.gemm(a, w, gelu=True, save_pre=True, aux=torch.empty(768, 768, device="cuda"))
    ref = torch.nn.functional.gelu(a.double() @ w.double().t())
    assert (out.double() - ref).abs().max() < 2e-5  # fast erf is fp32-exact to ~2e-7
    ab, wb = a.bfloat16(), w.bfloat16()
    outb = Kx.gemm(ab, wb)  # 36 tiles of 128^2 -> the 64^2-tile variant runs
    refb = ab.double() @ wb.double().t()
    assert (outb.double() - refb).abs().max() < 2e-2 * refb.abs().max()
    for a_km, b_km in ((False, True), (True, True), (True, False)):
        A = ab.t().contiguous() if a_km else ab
        Bm = wb.t().contiguous() if b_km else wb
        o = Kx.gemm(A, Bm, a_km=a_km, b_km=b_km, out_dtype=torch.float32)
        assert (o.double() - refb).abs().max() < 1e-3 * refb.abs().max(), (a_km, b_km)


def test_gemm_splitk_accumulate():
    from deepcoro_clip_amd import kernels as Kx

    g = torch.Generator(device="cpu").manual_seed(9)
    Mred, N, K = 5000, 96, 288  # weight-gradient shape: reduce over many rows
    dy = torch.randn(Mred, N, generator=g).to("cuda", torch.bfloat16)
    x = torch.randn(Mred, K, generator=g).to("cuda", torch.bfloat16)
    out = Kx.gemm(dy, x, a_km=True, b_km=True, out_dtype=torch.float32, splitk=8)
    ref = dy.float().t().double() @ x.float().double()
    assert (out.double() - ref).abs().max() < 1e-3 * ref.abs().max()


# ------------------------------------------------------------------ persistent LDS-DMA ring kernel (csrc/gemm_ring.hip)
# bf16 shapes with >= 192 tile-splits take the ring kernel; every case is checked with exact small-integer data
# first (any mis-placed chunk of the swizzled LDS images, any stale ring stage or mis-counted wait shows up as a wrong
# integer) and then with random data against fp64.
def _ring_case(M, N, K, a_km, b_km, out_dtype=torch.float32, splitk=1, seed=0, **kw):
    from deepcoro_clip_amd import kernels as Kx

    g = torch.Generator(device="cpu").manual_seed(seed + M + 3 * N + 7 * K)
    ai = torch.randint(-2, 3, (K, M) if a_km else (M, K), generator=g).to("cuda", torch.bfloat16)
    bi = torch.randint(-2, 3, (K, N) if b_km else (N, K), generator=g).to("cuda", torch.bfloat16)
    out = Kx.gemm(ai, bi, a_km=a_km, b_km=b_km, out_dtype=out_dtype, splitk=splitk, **kw)
    return out, _ref(ai, bi, a_km, b_km)


@pytest.mark.parametrize("a_km,b_km", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(3000, 1536, 384), (2049, 1160, 192), (25089, 96, 128), (1569, 3072, 64), (4096, 512, 1024),
                                   (25089, 288, 96), (6273, 576, 72), (3000, 1536, 200)])
def test_ring_layouts_exact(a_km, b_km, M, N, K):
    if (a_km and M % 8) or (b_km and N % 8):
        pytest.skip("k-major operands need 16-byte row chunks")
    out, ref = _ring_case(M, N, K, a_km, b_km)
    assert torch.equal(out.double(), ref)
    out16, _ = _ring_case(M, N, K, a_km, b_km, out_dtype=torch.bfloat16)
    assert torch.equal(out16.double(), ref.to(torch.bfloat16).double())


@pytest.mark.parametrize("splitk", [2, 7, 64])
@pytest.mark.parametrize("Kred,M,N", [(37656, 384, 1536), (9432, 768, 768), (150552, 96, 288), (5000, 1152, 384)])
def test_ring_splitk_wgrad_exact(splitk, Kred, M, N):
    """Weight-gradient shapes: both operands k-major, ragged K tail zero-filled by the buffer range check, fp32
    atomics, fused bias gradient (row sums of dY on the matrix core)."""
    from deepcoro_clip_amd import kernels as Kx

    g = torch.Generator(device="cpu").manual_seed(splitk + Kred)
    dy = torch.randint(-2, 3, (Kred, M), generator=g).to("cuda", torch.bfloat16)
    x = torch.randint(-2, 3, (Kred, N), generator=g).to("cuda", torch.bfloat16)
    db = torch.zeros(M, device="cuda")
    dw = torch.zeros(M, N, device="cuda")
    Kx.gemm(dy, x, a_km=True, b_km=True, out=dw, accumulate=True, splitk=splitk, colsum=db)
    assert torch.equal(dw.double(), dy.double().t() @ x.double())
    assert torch.equal(db.double(), dy.double().sum(0))
    Kx.gemm(dy, x, a_km=True, b_km=True, out=dw, accumulate=True, splitk=splitk, colsum=db)  # accumulates
    assert torch.equal(dw.double(), 2 * (dy.double().t() @ x.double()))


def test_ring_epilogues_and_random_data():
    from deepcoro_clip_amd import kernels as Kx

    g = torch.Generator(device="cpu").manual_seed(21)
    M, N, K = 5001, 1536, 384
    a = torch.randn(M, K, generator=g).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to("cuda", torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to("cuda", torch.bfloat16)
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = Kx.gemm(a, w, bias=bias, gelu=True, save_pre=True, aux=pre, resid=res)
    ref_pre = a.double() @ w.double().t() + bias.double()
    ref = torch.nn.functional.gelu(ref_pre) + res.double()
    assert (pre.double() - ref_pre).abs().max() < 2e-2 * ref_pre.abs().max()
    assert (out.double() - ref).abs().max() < 2e-2 * ref.abs().max()
    out2 = Kx.gemm(a, w, dgelu=True, aux=pre, alpha=0.5)
    x = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    ref2 = 0.5 * (a.double() @ w.double().t()) * x.grad
    assert (out2.double() - ref2).abs().max() < 2e-2 * ref2.abs().max()
    # strided output / residual views (a column slice of a wider buffer), bias only
    wide = torch.zeros(M, 2 * N, device="cuda", dtype=torch.bfloat16)
    Kx.gemm(a, w, bias=bias, out=wide[:, N:])
    assert (wide[:, N:].double() - ref_pre).abs().max() < 2e-2 * ref_pre.abs().max()
    assert wide[:, :N].abs().max() == 0


def test_ring_batched_and_matches_staged_kernel():
    """Batched form (patch embedding) and bit-identical results between the two GEMM kernels on the same inputs
    (same MFMA, same K order within a tile => identical fp32 sums when nothing is split)."""
    from deepcoro_clip_amd import kernels as Kx

    g = torch.Generator(device="cpu").manual_seed(4)
    Bn, L, Kp, Co = 24, 1568, 448, 96
    col = torch.randn(Bn * L, Kp, generator=g).to("cuda", torch.bfloat16)
    w = (torch.randn(Co, Kp, generator=g) * 0.05).to("cuda", torch.bfloat16)
    bias = torch.randn(Co, generator=g).cuda()
    out = torch.zeros(Bn, L + 1, Co, device="cuda", dtype=torch.bfloat16)
    Kx.gemm_batched(col, w, out[:, 1:], M=L, N=Co, K=Kp, lda=Kp, ldb=Kp, ldc=Co, batch=Bn, sA=L * Kp, sB=0,
                    sC=(L + 1) * Co, bias=bias)
    ref = (col.double() @ w.double().t() + bias.double()).view(Bn, L, Co)
    assert (out[:, 1:].double() - ref).abs().max() < 2e-2 * ref.abs().max()
    assert out[:, 0].abs().max() == 0
    a = torch.randn(6000, 768, generator=g).to("cuda", torch.bfloat16)
    b = torch.randn(1152, 768, generator=g).to("cuda", torch.bfloat16)
    o_ring = Kx.gemm(a, b, out_dtype=torch.float32)
    old = Kx.GEMM_VARIANT
    try:
        Kx.GEMM_VARIANT = 3 << 8  # force the register-staged kernel
        o_staged = Kx.gemm(a, b, out_dtype=torch.float32)
    finally:
        Kx.GEMM_VARIANT = old
    assert torch.equal(o_ring, o_staged)


@pytest.mark.parametrize("b_km", [False, True])
@pytest.mark.parametrize("M,N,K", [(1, 8, 64), (7, 24, 100), (24, 512, 2048), (24, 2048, 512), (24, 24, 512), (32, 520, 1536),
                                   (17, 1536, 512)])
def test_gemm_skinny_fp32_rows(b_km, M, N, K):
    """fp32 GEMMs with a handful of rows (aggregator / projection heads on [batch, 512]) take the K-over-threads kernel:
    exact on integer data, every epilogue against the torch expression."""
    from deepcoro_clip_amd import kernels as Kx

    g = torch.Generator(device="cpu").manual_seed(M * 13 + N + K)
    ai = torch.randint(-3, 4, (M, K), generator=g).float().cuda()
    bi = torch.randint(-3, 4, (K, N) if b_km else (N, K), generator=g).float().cuda()
    outi = Kx.gemm(ai, bi, b_km=b_km)
    assert torch.equal(outi.double(), _ref(ai, bi, False, b_km)), "exact integer data"
    a = torch.randn(M, K, generator=g).cuda()
    b = (torch.randn((K, N) if b_km else (N, K), generator=g) * 0.05).cuda()
    bias, resid = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    pre = a.double() @ (b.double() if b_km else b.double().t()) + bias.double()
    aux = torch.empty(M, N, device="cuda")
    out = Kx.gemm(a, b, b_km=b_km, bias=bias, gelu=True, save_pre=True, aux=aux)
    assert (aux.double() - pre).abs().max() < 1e-4
    assert (out.double() - torch.nn.functional.gelu(pre)).abs().max() < 1e-4
    out = Kx.gemm(a, b, b_km=b_km, bias=bias, resid=resid)
    assert (out.double() - (pre + resid.double())).abs().max() < 1e-4
    x = torch.randn(M, N, generator=g).cuda().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    out = Kx.gemm(a, b, b_km=b_km, dgelu=True, aux=x.detach())
    assert (out.double() - (pre - bias.double()) * x.grad.double()).abs().max() < 1e-4
    # same numbers as the tiled kernel (variant 3 switches the skinny path off)
    Kx.GEMM_VARIANT, keep = 3 << 8, Kx.GEMM_VARIANT
    try:
        tiled = Kx.gemm(a, b, b_km=b_km, bias=bias, resid=resid)
    finally:
        Kx.GEMM_VARIANT = keep
    assert (tiled - Kx.gemm(a, b, b_km=b_km, bias=bias, resid=resid)).abs().max() < 1e-4
