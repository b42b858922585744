"""GPU numerics: snag_gemm_f32 (exact-f32 MFMA) vs a plain PyTorch fp32/fp64 reference of the same op."""
import math

import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(a, b, a_kc, b_kc):
    A = a.double() if a_kc else a.double().t()
    B = b.double() if b_kc else b.double().t()
    return A @ B.t()


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (45000, 384, 128), (257, 300, 300), (1000, 128, 4096),
                                   (64, 128, 16), (5, 3, 7), (129, 130, 33), (300, 400, 1200), (96, 128, 1)])
@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
def test_gemm_layouts(M, N, K, a_kc, b_kc):
    from snag_amd import ops
    if M * K > 20_000_000 and not (a_kc and b_kc):
        pytest.skip("large case only in the NT layout")
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((M, K) if a_kc else (K, M), generator=g)
    b = torch.randn((N, K) if b_kc else (K, N), generator=g)
    c = ops.gemm(a.to(DEV), b.to(DEV), a_kcontig=a_kc, b_kcontig=b_kc)
    ref = _ref(a, b, a_kc, b_kc)
    # fp32 accumulation error bound ~ K * eps * |a||b| ; compare against fp64
    assert_close(c, ref, atol=2e-6 * K + 1e-5, rtol=2e-5, what=f"gemm {M}x{N}x{K}")


def test_gemm_exact_integers_asymmetric():
    """Small-integer operands are exact in fp32: catches any row/col or k-mapping swap bit-for-bit."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-4, 5, (200, 70), generator=g).float()
    b = torch.randint(-4, 5, (150, 70), generator=g).float()
    c = ops.gemm(a.to(DEV), b.to(DEV))
    assert torch.equal(c.cpu(), a @ b.t())
    c2 = ops.gemm(a.t().contiguous().to(DEV), b.t().contiguous().to(DEV), a_kcontig=False, b_kcontig=False)
    assert torch.equal(c2.cpu(), a @ b.t())


def test_gemm_epilogue_chain():
    from snag_amd import ops
    g = torch.Generator().manual_seed(11)
    M, N, K = 333, 128, 200
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    bias, r1, r2 = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    keep = (torch.rand(M, N, generator=g) >= 0.1).to(torch.uint8)
    for act, fn in (("gelu", lambda x: 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))), ("relu", torch.relu), (None, lambda x: x)):
        c, pre = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), alpha=0.5, act=act, want_pre_act=True,
                          keep_mask=keep.to(DEV), keep_p=0.1, add1=r1.to(DEV), add2=r2.to(DEV))
        u = 0.5 * (a.double() @ b.double().t()) + bias.double()
        ref = fn(u) * keep.double() / 0.9 + r1.double() + r2.double()
        assert_close(pre, u, atol=1e-5, what="pre_act")
        assert_close(c, ref, atol=2e-5, what=f"epilogue {act}")
    # accumulate + strided output (writes one modality slot of a stacked [M, 3, N] tensor)
    stacked = torch.zeros(M, 3, N, device=DEV)
    ops.gemm(a.to(DEV), b.to(DEV), out=stacked[:, 1, :])
    ops.gemm(a.to(DEV), b.to(DEV), out=stacked[:, 1, :], accumulate=True)
    assert_close(stacked[:, 1, :], 2 * (a.double() @ b.double().t()), atol=5e-5)
    assert float(stacked[:, 0, :].abs().max()) == 0 and float(stacked[:, 2, :].abs().max()) == 0


@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (False, False)])
@pytest.mark.parametrize("splitk", [True, False])
def test_gemm_k_zero_applies_epilogue_only(a_kc, b_kc, splitk):
    """K = 0 is legal (gemm_check admits it): C = epilogue(0), so bias / add1 / accumulate still apply.  The dispatcher's split
    arithmetic once divided by zero here (host SIGFPE)."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(2)
    M, N = 70, 132
    a = torch.empty((M, 0) if a_kc else (0, M), device=DEV)
    b = torch.empty((N, 0) if b_kc else (0, N), device=DEV)
    bias, r1 = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    c = ops.gemm(a, b, a_kcontig=a_kc, b_kcontig=b_kc, bias=bias, add1=r1, splitk=splitk)
    assert torch.equal(c, bias[None, :] + r1)
    out = torch.full((M, N), 2.0, device=DEV)
    ops.gemm(a, b, a_kcontig=a_kc, b_kcontig=b_kc, out=out, accumulate=True, splitk=splitk)
    assert torch.equal(out, torch.full((M, N), 2.0, device=DEV))


def test_gemm_splitk_long_k_deterministic():
    """dW = dY^T X shape: tiny output, long reduction; split-K slabs are summed in a fixed order."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(5)
    Kred, Nout, Kin = 45000, 128, 128
    dy, x = torch.randn(Kred, Nout, generator=g), torch.randn(Kred, Kin, generator=g)
    dyd, xd = dy.to(DEV), x.to(DEV)
    c1 = ops.gemm(dyd, xd, a_kcontig=False, b_kcontig=False)
    c2 = ops.gemm(dyd, xd, a_kcontig=False, b_kcontig=False)
    assert torch.equal(c1, c2)
    assert_close(c1, dy.double().t() @ x.double(), atol=8e-3, rtol=1e-5)
    c3 = ops.gemm(dyd, xd, a_kcontig=False, b_kcontig=False, splitk=False)
    assert_close(c3, c1, atol=3e-2)      # one 45,000-long fp32 chain vs 86 short ones


@pytest.mark.parametrize("n,k,pad", [(7, 3, 0), (129, 16, 0), (300, 37, 3), (301, 300, 0), (1000, 300, 4), (64, 0, 0)])
def test_gram_is_symmetric_and_matches_gemm(n, k, pad):
    """snag_gram_f32: upper-triangle tiles + transposed stores.  S == S^T bit for bit; equals the plain GEMM's values."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(n * 31 + k)
    zfull = torch.randn(n, k + pad, generator=g).to(DEV)
    z = zfull[:, :k] if k else torch.empty(n, 0, device=DEV)
    s = ops.gram(z, alpha=0.25)
    assert torch.equal(s, s.t().contiguous())
    if k:
        assert_close(s, 0.25 * (z.double().cpu() @ z.double().cpu().t()), atol=1e-5 * max(1.0, k ** 0.5))
        full = ops.gemm(z, z, alpha=0.25)
        assert torch.equal(torch.triu(s), torch.triu(full))
    else:
        assert float(s.abs().max()) == 0.0
