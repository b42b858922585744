"""GPU parity of the two transformer-free pieces of the MKGC read-out family: joint_way "atten_weight" (README.md:51; arithmetic pinned to
the reference's EVA_tools.SelfAttention by golden G15) and the README's POOL=1 feature pooling (README.md:68,79; unpinned: the oracle's
windows are torch's adaptive average pool)."""
import pytest
import torch

from helpers import T, assert_close, load
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("tag", ["m3_d128", "m4_d300", "m6_d64"])
def test_atten_weight_golden(tag):
    """snag_atten_weight_fwd / _bwd against what the reference's SelfAttention.forward (model/EVA_tools.py:203-232) and its autograd produced."""
    from snag_amd import ops
    z = load("g15_atten_weight.npz")
    x, a, cw = T(z[f"{tag}.x"], DEV), T(z[f"{tag}.a"], DEV), T(z[f"{tag}.cw"], DEV)
    rep, aw = ops.atten_weight_fwd(x, a)
    M = x.shape[1]
    assert_close(rep, z[f"{tag}.rep"], atol=1e-5, rtol=1e-5, what="rep")
    assert_close(aw[:, :M], z[f"{tag}.w"], atol=2e-6, rtol=1e-5, what="weights")
    dx, da = ops.atten_weight_bwd(x, a, aw, cw)
    assert_close(dx, z[f"{tag}.g.x"], atol=2e-5, rtol=1e-4, what="g.x")
    assert_close(da, z[f"{tag}.g.a"], atol=2e-4, rtol=1e-4, what="g.a")


@pytest.mark.parametrize("rows,M,D", [(1, 1, 4), (15, 2, 8), (17, 3, 128), (1000, 3, 128), (4099, 5, 260), (15000, 3, 128), (9000, 6, 512)])
def test_atten_weight_vs_oracle(rows, M, D):
    """Ragged row counts (the last block's idle groups), every token count, widths on both register plans; float64 oracle."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(rows + 31 * M + D)
    x = torch.randn(rows, M, D, generator=g)
    a = torch.randn(D, generator=g) * (2.0 / D ** 0.5)
    cw = torch.randn(rows, D, generator=g)
    rep, aw = ops.atten_weight_fwd(x.to(DEV), a.to(DEV))
    xd, ad = x.double().requires_grad_(True), a.double().requires_grad_(True)
    ref = O.atten_weight_readout(xd, ad)
    (ref * cw.double()).sum().backward()
    assert_close(rep, ref, atol=1e-5, rtol=1e-5, what="rep")
    # a row-strided gradient (the engine's is dense; a [rows, 2 D] buffer's left half exercises ld_dout)
    wide = torch.zeros(rows, 2 * D, device=DEV)
    wide[:, :D] = cw.to(DEV)
    dx, da = ops.atten_weight_bwd(x.to(DEV), a.to(DEV), aw, wide[:, :D])
    assert_close(dx, xd.grad, atol=2e-5, rtol=1e-4, what="d_x")
    assert_close(da, ad.grad, atol=2e-4 * max(1.0, float(ad.grad.abs().max())), rtol=1e-4, what="d_a")
    # fixed-order partial slabs: the same bits on a second run; accumulation adds
    dx2, da2 = ops.atten_weight_bwd(x.to(DEV), a.to(DEV), aw, wide[:, :D])
    assert torch.equal(da, da2) and torch.equal(dx, dx2)
    _, da3 = ops.atten_weight_bwd(x.to(DEV), a.to(DEV), aw, wide[:, :D], d_a=da2.clone(), accumulate_da=True)
    assert_close(da3, 2 * da, atol=1e-6 * max(1.0, float(da.abs().max())), rtol=1e-6, what="accumulate")


def test_atten_weight_empty_and_refusals():
    from snag_amd import ops, _lib
    x = torch.zeros(0, 3, 128, device=DEV)
    a = torch.ones(128, device=DEV)
    rep, aw = ops.atten_weight_fwd(x, a)
    assert rep.shape == (0, 128)
    _, da = ops.atten_weight_bwd(x, a, aw, torch.zeros(0, 128, device=DEV))
    assert float(da.abs().max()) == 0.0
    with pytest.raises(_lib.SnagHipError):
        ops.atten_weight_fwd(torch.zeros(4, 3, 6, device=DEV), torch.ones(6, device=DEV))        # 4 does not divide D
    with pytest.raises(_lib.SnagHipError):
        ops.atten_weight_fwd(torch.zeros(4, 7, 8, device=DEV), torch.ones(8, device=DEV))        # more than 6 tokens
    # rows travel as float4: a base pointer off the 16-byte grid or a row stride that is not a multiple of four floats is refused --
    # by the wrapper (ValueError) and by the C ABI itself (SNAG_ERR_UNSUPPORTED = -2), never read misaligned
    lib, st = _lib.load(), _lib.stream()
    x, a = torch.zeros(4, 3, 8, device=DEV), torch.ones(16, device=DEV)
    odd = torch.zeros(4, 9, device=DEV)[:, :8]                   # ld_out = 9
    off = torch.zeros(4 * 3 * 8 + 1, device=DEV)[1:].view(4, 3, 8)          # base 4 bytes off
    aw = torch.zeros(4, 6, device=DEV)
    with pytest.raises(ValueError):
        ops.atten_weight_fwd(x, a[:8], out=odd)
    with pytest.raises(ValueError):
        ops.atten_weight_fwd(off, a[:8])
    with pytest.raises(ValueError):
        ops.atten_weight_bwd(x, a[:8], aw, odd)
    out = torch.zeros(4, 8, device=DEV)
    assert lib.snag_atten_weight_fwd(x.data_ptr(), 4, 3, 8, a.data_ptr(), odd.data_ptr(), 9, aw.data_ptr(), st) == -2
    assert lib.snag_atten_weight_fwd(off.data_ptr(), 4, 3, 8, a.data_ptr(), out.data_ptr(), 8, aw.data_ptr(), st) == -2
    assert lib.snag_atten_weight_fwd(x.data_ptr(), 4, 3, 8, a.data_ptr() + 4, out.data_ptr(), 8, aw.data_ptr(), st) == -2
    work = torch.zeros(max(1, lib.snag_atten_weight_work_floats(8)), device=DEV)
    da = torch.zeros(8, device=DEV)
    assert lib.snag_atten_weight_bwd(x.data_ptr(), 4, 3, 8, a.data_ptr(), aw.data_ptr(), odd.data_ptr(), 9, torch.zeros_like(x).data_ptr(), da.data_ptr(), 0,
                                     work.data_ptr(), st) == -2
    assert lib.snag_atten_weight_bwd(x.data_ptr(), 4, 3, 8, a.data_ptr(), aw.data_ptr(), out.data_ptr(), 8, off.data_ptr(), da.data_ptr(), 0,
                                     work.data_ptr(), st) == -2
    assert lib.snag_atten_weight_fwd(x.data_ptr(), 4, 3, 8, a.data_ptr(), out.data_ptr(), 8, aw.data_ptr(), st) == 0


@pytest.mark.parametrize("N,view,out", [(37, (1, 4096), (1, 256)), (37, (64, 64), (16, 16)), (20, (12, 64), (4, 64)), (9, (1, 770), (1, 256)), (5, (1, 100), (1, 100)),
                                         (3, (7, 11), (3, 5)), (15000, (1, 768), (1, 256))])
def test_adaptive_avg_pool_rows_vs_oracle(N, view, out):
    from snag_amd import ops
    g = torch.Generator().manual_seed(N)
    x = torch.randn(N, view[0] * view[1], generator=g)
    got = ops.adaptive_avg_pool_rows(x.to(DEV), out, view_hw=view)
    ref = O.adaptive_avg_pool_rows(x, out, view_hw=view)
    assert_close(got, ref, atol=2e-6, rtol=1e-6, what="pooled")
    # a row-strided source (columns of a wider table) gives the same rows
    wide = torch.zeros(N, x.shape[1] + 8, device=DEV)
    wide[:, :x.shape[1]] = x.to(DEV)
    assert torch.equal(ops.adaptive_avg_pool_rows(wide[:, :x.shape[1]], out, view_hw=view), got)


@pytest.mark.parametrize("views", [None, [(64, 64, 16, 16), (12, 64, 4, 64)]])
def test_engine_with_use_pool_equals_oracle_on_pooled_features(views):
    """MKGCConfig.use_pool (README POOL=1): the engine pools the raw tables once at load and trains on the pooled ones -- step vs the oracle fed
    oracle-pooled features; the engine's configuration records the pooled widths and where they came from."""
    from snag_amd.mkgc import MKGCConfig, MKGCHotPath
    from test_mkgc_gpu import oracle_params
    N, R, D, B, K = 600, 9, 128, 128, 32
    cfg = MKGCConfig(n_ent=N, n_rel=R, dim=D, feat_dims=[4096, 768], num_proj=2, margin=6.0, neg_num=K, dropout=0.0, add_noise=0, seed=5,
                     use_pool=1, pool_dim=256, pool_views=views)
    g = torch.Generator().manual_seed(8)
    feats = [torch.rand(N, 4096, generator=g), torch.randn(N, 768, generator=g)]
    eng = MKGCHotPath(cfg, feats, device=DEV)
    assert eng.cfg.feat_dims == [256, 256] and eng.pooled_from == [4096, 768] and eng.cfg.use_pool == 0
    assert eng.state_dict()["pooled_from"] == [4096, 768]
    pooled = [O.adaptive_avg_pool_rows(f, (v[2], v[3]), view_hw=(v[0], v[1])) if v else O.adaptive_avg_pool_rows(f, 256) for f, v in zip(feats, views or [None, None])]
    for got, ref in zip(eng.feats, pooled):
        assert_close(got, ref, atol=2e-6, rtol=1e-6, what="pooled table")
    eng.p["ent"].copy_(torch.randn(N, D, generator=g).to(DEV))
    idx = [torch.randint(0, hi, sz, generator=g, dtype=torch.int32) for hi, sz in ((N, (B,)), (R, (B,)), (N, (B,)), (N, (B, K)))]
    loss, pos, ng = eng.forward(*[i.to(DEV) for i in idx], want_scores=True, with_backward=True)
    eng.backward()
    p, proj, fp = oracle_params(eng)
    oloss, opos, oneg, _ = O.mkgc_forward(pooled, proj, p["ent"], fp, p["rel"], *idx, num_heads=cfg.num_heads, margin=cfg.margin)
    oloss.backward()
    assert_close(pos, opos, atol=1e-4, what="pos")
    assert_close(ng, oneg, atol=1e-4, what="neg")
    grads = eng.named_gradients()
    for k in ("proj0.l1", "proj1.l1", "ent", "layer0.wqkv"):
        scale = max(1.0, float(p[k].grad.abs().max()))
        assert_close(grads[k], p[k].grad, atol=2e-4 * scale, rtol=1e-4, what=f"grad {k}")


def test_atten_weight_engine_trains_and_checkpoints(tmp_path):
    """A few optimiser steps reduce the loss through the attention vector too, and a checkpoint round-trips ``atten.a``."""
    from snag_amd.mkgc import MKGCConfig, MKGCHotPath
    N, R, D, B, K = 500, 7, 64, 128, 16
    cfg = MKGCConfig(n_ent=N, n_rel=R, dim=D, feat_dims=[80, 40], joint_way="atten_weight", neg_num=K, lr=1e-2, seed=4)
    g = torch.Generator().manual_seed(6)
    feats = [torch.rand(N, 80, generator=g), torch.randn(N, 40, generator=g)]
    eng = MKGCHotPath(cfg, feats, device=DEV)
    assert "atten.a" in eng.named_parameters() and not any(k.startswith("layer") for k in eng.named_parameters())
    idx = [torch.randint(0, hi, sz, generator=g, dtype=torch.int32).to(DEV) for hi, sz in ((N, (B,)), (R, (B,)), (N, (B,)), (N, (B, K)))]
    a0 = eng.p["atten.a"].clone()
    eng.update_noise()
    first = float(eng.train_step(*idx))
    for _ in range(30):
        last = float(eng.train_step(*idx))
    assert last < first
    assert not torch.equal(a0, eng.p["atten.a"])
    path = eng.save_checkpoint(str(tmp_path / "aw.pt"))
    eng2 = MKGCHotPath(cfg, feats, device=DEV).load_checkpoint(path)
    assert torch.equal(eng2.p["atten.a"], eng.p["atten.a"])
    res = eng.link_prediction(*idx[:3])
    assert res["rank"].shape == (B,)
