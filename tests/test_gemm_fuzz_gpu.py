"""GPU: seeded random sweep of snag_gemm_f32 -- shapes, layouts, row pitches, epilogue combinations -- against an fp64 product.
Covers the dispatcher's corners: 64x128 / 128x128 / 128x160 tiles, split-K, stream-K, checked vs LDS-DMA loaders, vector vs scalar epilogue."""
import math
import random

import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(rng, big):
    if big:
        M = rng.choice([8192, 9000, 15000, 20001])
        N = rng.choice([64, 128, 160, 300, 384])
        K = rng.choice([300, 1000, 2048, 4100])
    else:
        M, N, K = rng.randint(1, 400), rng.randint(1, 400), rng.randint(0, 300)
    akc, bkc = rng.random() < 0.5, rng.random() < 0.5
    pad = lambda: rng.choice([0, 0, 1, 3, 4, 8, 32])
    return M, N, K, akc, bkc, pad(), pad(), pad()


@pytest.mark.parametrize("seed,big", [(s, False) for s in range(12)] + [(100 + s, True) for s in range(4)])
def test_gemm_random_cases(seed, big):
    from snag_amd import ops
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    for _ in range(2 if big else 12):
        M, N, K, akc, bkc, pa, pb, pc = _case(rng, big)
        a_shape = (M, K) if akc else (K, M)
        b_shape = (N, K) if bkc else (K, N)
        a = torch.randn(a_shape[0], a_shape[1] + pa, generator=g).to(DEV)[:, :a_shape[1]]
        b = torch.randn(b_shape[0], b_shape[1] + pb, generator=g).to(DEV)[:, :b_shape[1]]
        out = torch.full((M, N + pc), 3.0, device=DEV)[:, :N]
        kw = {}
        use_bias = rng.random() < 0.4
        bias = torch.randn(N, generator=g) if use_bias else None
        act = rng.choice([None, None, "relu", "gelu"])
        add1 = torch.randn(M, N, generator=g) if rng.random() < 0.3 else None
        keep = (torch.rand(M, N, generator=g) >= 0.2).to(torch.uint8) if rng.random() < 0.3 else None
        accumulate = rng.random() < 0.2
        alpha = rng.choice([1.0, 0.5, -2.0])
        if bias is not None:
            kw["bias"] = bias.to(DEV)
        if add1 is not None:
            kw["add1"] = add1.to(DEV)
        if keep is not None:
            kw["keep_mask"], kw["keep_p"] = keep.to(DEV), 0.2
        A = (a if akc else a.t()).double().cpu()
        B = (b if bkc else b.t()).double().cpu()
        u = alpha * (A @ B.t())
        if bias is not None:
            u = u + bias.double()
        if act == "relu":
            u = torch.relu(u)
        elif act == "gelu":
            u = 0.5 * u * (1 + torch.erf(u / math.sqrt(2)))
        if keep is not None:
            u = u * keep.double() / 0.8
        if add1 is not None:
            u = u + add1.double()
        if accumulate:
            u = u + 3.0
        ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc, alpha=alpha, act=act, out=out, accumulate=accumulate, **kw)
        tol = 2e-6 * max(1.0, math.sqrt(K)) * max(1.0, float(u.abs().max()))
        assert_close(out, u, atol=tol, what=f"M={M} N={N} K={K} akc={akc} bkc={bkc} pads={pa},{pb},{pc} bias={use_bias} act={act} "
                                             f"add1={add1 is not None} keep={keep is not None} acc={accumulate}")
        if pc:
            full = torch.as_strided(out, (M, N + pc), (N + pc, 1))
            assert float((full[:, N:] - 3.0).abs().max()) == 0.0, "the GEMM wrote into the row padding"
