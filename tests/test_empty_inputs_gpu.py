"""Empty and degenerate inputs through the C-ABI: zero triples, zero rows, zero negatives, one-row / one-column matrices.  Every entry
either does nothing and returns SNAG_OK or refuses with a clean error -- none may fault or leave an error state behind."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tables(n=50, r=4, d=128, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, d, generator=g).to(DEV), torch.randn(r, d, generator=g).to(DEV)


def _ok_after():
    """the library still works after the degenerate call"""
    from snag_amd import ops
    x = torch.arange(10, dtype=torch.float32, device=DEV)
    assert float(ops.reduce_sum(x)) == 45.0


def test_scorer_with_no_triples_or_no_negatives():
    from snag_amd import ops
    ent, rel = _tables()
    e = lambda *s: torch.zeros(*s, dtype=torch.int32, device=DEV)
    for fn in (ops.score_margin_fwd_bwd, ops.score_margin_fwd_bwd_seg):
        out = fn(ent, rel, e(0), e(0), e(0), e(0, 4), 1.0)
        torch.cuda.synchronize()
        for o in out:
            if torch.is_tensor(o) and o.numel():
                assert torch.isfinite(o).all()
    de, dr = ops.score_margin_bwd(ent, rel, e(0), e(0), e(0), e(0, 4), 1.0)
    assert float(de.abs().sum()) == 0.0 and float(dr.abs().sum()) == 0.0
    # K = 0: positives only, the hinge has nothing to rank against -> zero loss, zero gradient
    h = torch.tensor([1, 2, 3], dtype=torch.int32, device=DEV)
    pos, ng, lp = ops.score_margin_fwd(ent, rel, h, h % 4, (h + 1) % 50, e(3, 0), 1.0)
    assert pos.shape == (3,) and ng.shape == (3, 0) and float(lp.abs().sum()) == 0.0
    _ok_after()


def test_sampler_with_no_triples():
    from snag_amd import ops
    e = torch.zeros(0, dtype=torch.int32, device=DEV)
    neg = ops.sample_negatives(e, e, e, 8, 100, 5, seed=1)
    assert neg.shape == (0, 8)
    _ok_after()


@pytest.mark.parametrize("M,N,K", [(0, 128, 64), (128, 0, 64), (1, 1, 1), (1, 128, 4096), (4096, 1, 5)])
def test_gemm_degenerate_shapes(M, N, K):
    from snag_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a, b = torch.randn(M, K, generator=g).to(DEV), torch.randn(N, K, generator=g).to(DEV)
    c = ops.gemm(a, b)
    assert c.shape == (M, N)
    if M and N:
        torch.testing.assert_close(c.double(), a.double() @ b.double().t(), atol=2e-6 * K + 1e-5, rtol=2e-5)
    _ok_after()


def test_gram_of_nothing_and_of_one_row():
    from snag_amd import ops
    z = torch.randn(1, 300, device=DEV)
    s = ops.gram(z)
    torch.testing.assert_close(s.double(), z.double() @ z.double().t(), atol=1e-4, rtol=1e-5)
    assert ops.gram(torch.zeros(0, 300, device=DEV)).shape == (0, 0)
    _ok_after()


def test_reductions_of_nothing():
    from snag_amd import ops
    assert float(ops.reduce_sum(torch.zeros(0, device=DEV))) == 0.0
    assert float(ops.colsum(torch.zeros(0, 7, device=DEV)).abs().sum()) == 0.0
    _ok_after()


def test_fusion_over_no_entities():
    from types import SimpleNamespace
    from snag_amd.fusion import MformerFusion
    args = SimpleNamespace(hidden_size=128, num_attention_heads=2, intermediate_size=400, num_hidden_layers=1, use_intermediate=0)
    m = MformerFusion(args, modal_num=3).to(DEV).eval()
    out = m([torch.zeros(0, 128, device=DEV) for _ in range(3)])
    torch.cuda.synchronize()
    for o in out:
        if torch.is_tensor(o):
            assert o.shape[0] == 0
    # one entity, one token: softmax over a single key
    one = m([torch.randn(1, 128, device=DEV)])
    torch.cuda.synchronize()
    assert all(torch.isfinite(o).all() for o in one if torch.is_tensor(o))
    _ok_after()


def test_similarity_eval_of_one_pair():
    from snag_amd import evalsim
    g = torch.Generator().manual_seed(5)
    l, r = torch.randn(1, 64, generator=g).to(DEV), torch.randn(1, 64, generator=g).to(DEV)
    res = evalsim.rank_metrics(l, r, csls_k=1, top_k=[1])
    torch.cuda.synchronize()
    assert res is not None
    _ok_after()


def test_noise_and_dropout_over_nothing():
    from snag_amd import ops
    mean, std = torch.zeros(16, device=DEV), torch.ones(16, device=DEV)
    out = ops.noise_apply(torch.zeros(0, 16, device=DEV), mean, std, torch.zeros(0, dtype=torch.uint8, device=DEV), 0.7, seed=3)
    assert out.shape == (0, 16)
    assert ops.dropout_keep_mask((0,), 0.1, seed=5).numel() == 0
    assert ops.gaussian_fill(mean, std, 0, seed=7).shape == (0, 16)
    _ok_after()


def test_infonce_with_one_pair_and_with_none():
    """B = 1: each row's only in-batch candidate besides itself is its own target -> loss 0, gradient 0; B = 0 is refused."""
    from snag_amd.losses import icl_loss
    emb = torch.randn(6, 32, device=DEV, requires_grad=True)
    crit = icl_loss(tau=0.1)
    loss = crit(emb, torch.tensor([[1, 4]], device=DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss.detach())) < 1e-6 and float(emb.grad.abs().max()) < 1e-6
    with pytest.raises(ValueError):
        crit(emb, torch.zeros(0, 2, dtype=torch.int64, device=DEV))
    _ok_after()


def test_evaluation_over_nothing_is_refused():
    from snag_amd import evalsim
    ent, rel = _tables()
    e = torch.zeros(0, dtype=torch.int32, device=DEV)
    with pytest.raises(ValueError):
        evalsim.link_prediction(ent, rel, e, e, e)
    with pytest.raises(ValueError):
        evalsim.rank_metrics(torch.zeros(0, 16, device=DEV), torch.zeros(0, 16, device=DEV))
    _ok_after()
