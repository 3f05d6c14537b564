"""Crew variant of the wide-tile GEMM (persistent 256 x 128 tiles, storer waves behind an LDS mailbox): exact integer data
(every product and partial sum representable, so any mis-addressed tile, lost K-step or mailbox race shows as a wrong integer),
all four layouts, ragged edges, and the fused epilogues against the ring kernel."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ints(shape, g, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, generator=g).float()


@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (512, 256, 96), (1000, 264, 200), (50208, 1536, 384), (2056, 1152, 768),
                                   (130 * 8, 8, 64), (8192, 384, 1536)])
@pytest.mark.parametrize("a_km,b_km", [(False, False), (False, True), (True, False), (True, True)])
def test_crew_gemm_exact_integers(M, N, K, a_km, b_km):
    from deepcoro_clip_amd import kernels as K_

    if M * N > 40e6 and (a_km or b_km) and not (not a_km and b_km):
        pytest.skip("the large shape runs in the two layouts the model uses")
    g = torch.Generator().manual_seed(M + N + K)
    A = _ints((M, K), g, -2, 3)
    B = _ints((N, K), g, -2, 3)
    ref = (A.double() @ B.double().T)
    a = (A.T.contiguous() if a_km else A).bfloat16().cuda()
    b = (B.T.contiguous() if b_km else B).bfloat16().cuda()
    old = K_.GEMM_VARIANT
    K_.GEMM_VARIANT = 12 << 8
    try:
        out = K_.gemm(a, b, a_km=a_km, b_km=b_km)
    finally:
        K_.GEMM_VARIANT = old
    want = ref.to(torch.bfloat16).float()  # |sums| <= 4 K: exact in fp32, rounded once to bf16
    assert torch.equal(out.float().cpu(), want)


@pytest.mark.parametrize("epi", ["bias", "bias_resid", "gelu_pre", "dgelu"])
def test_crew_gemm_epilogues_match_ring_kernel(epi):
    from deepcoro_clip_amd import kernels as K_

    M, N, Kd = 3000, 1536, 384
    g = torch.Generator().manual_seed(3)
    a = (torch.rand(M, Kd, generator=g) * 2 - 1).bfloat16().cuda()
    b_km = epi == "dgelu"
    b = (torch.rand((Kd, N) if b_km else (N, Kd), generator=g) * 2 - 1).bfloat16().cuda()
    kw = {}
    if "bias" in epi or epi == "gelu_pre":
        kw["bias"] = torch.randn(N, generator=g).cuda()
    if epi == "bias_resid":
        kw["resid"] = torch.randn(M, N, generator=g).bfloat16().cuda()

    def run(variant):
        old = K_.GEMM_VARIANT
        K_.GEMM_VARIANT = variant << 8
        try:
            args = dict(kw)
            aux = None
            if epi == "gelu_pre":
                aux = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
                args.update(gelu=True, save_pre=True, aux=aux)
            if epi == "dgelu":
                aux = torch.randn(M, N, generator=torch.Generator().manual_seed(9)).bfloat16().cuda()
                args.update(dgelu=True, aux=aux)
            return K_.gemm(a, b, b_km=b_km, **args), aux
        finally:
            K_.GEMM_VARIANT = old

    (o1, x1), (o0, x0) = run(12), run(3)   # 3 = the tiled kernel (no ring, no wide)
    assert (o1.float() - o0.float()).abs().max().item() <= 2.0 ** -6 * max(1.0, o0.float().abs().max().item())
    if epi == "gelu_pre":
        assert (x1.float() - x0.float()).abs().max().item() <= 2.0 ** -6 * max(1.0, x0.float().abs().max().item())
