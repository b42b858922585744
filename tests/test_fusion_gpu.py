"""GPU parity: MformerFusion on HIP (through the C ABI) vs golden G1 (reference MformerFusion.forward + autograd)."""
import types

import numpy as np
import pytest
import torch

from helpers import T, assert_close, fusion_cases, fusion_inputs, fusion_masks, load

pytestmark = pytest.mark.gpu
DEV = "cuda"
OUTS = ("joint", "joint_fz", "hidden", "weight_norm")


def build_module(z, cfg, params):
    from snag_amd.fusion import MformerFusion
    args = types.SimpleNamespace(hidden_size=cfg["D"], num_attention_heads=cfg["H"], num_hidden_layers=cfg["L"],
                                 use_intermediate=int(cfg["ffn"]), intermediate_size=cfg["I"])
    mod = MformerFusion(args, modal_num=cfg["M"])
    missing, unexpected = mod.load_state_dict({k: v.cpu() for k, v in params.items()}, strict=True), None
    return mod.to(DEV)


@pytest.mark.parametrize("case", fusion_cases())
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_fusion_golden(case, mode):
    z = load(case)
    embs, params, cfg = fusion_inputs(z)
    mod = build_module(z, cfg, params)
    mod.train(mode == "train")
    embs_d = [e.to(DEV).requires_grad_(True) if e is not None else None for e in embs]
    masks = fusion_masks(z, cfg["L"], DEV) if mode == "train" else None
    outs = mod(embs_d, drop_masks=masks)
    for nm, o in zip(OUTS, outs):
        assert_close(o, z[f"{mode}.{nm}"], atol=2e-5, rtol=1e-5, what=f"{case}/{mode}/{nm}")
    scalar = sum((o * T(z[f"cw{i}"], DEV)).sum() for i, o in enumerate(outs))
    scalar.backward()
    for i, e in enumerate(embs_d):
        if e is not None:
            assert_close(e.grad, z[f"{mode}.g.emb{i}"], atol=1e-4, rtol=1e-4, what=f"{case}/{mode}/g.emb{i}")
    sd = dict(mod.named_parameters())
    checked = 0
    for k, p in sd.items():
        key = f"{mode}.g.p.{k}"
        if key in z:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            ref = z[key]
            tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
            assert_close(g, ref, atol=tol, rtol=1e-4, what=key)
            checked += 1
    assert checked >= 1


def test_state_dict_keys_match_reference():
    """SURVEY.md par. 8b: checkpoint key names (probed from the reference) are preserved."""
    z = load("g1_fusion_m4_d128_h2_ffn.npz")
    _, params, cfg = fusion_inputs(z)
    from snag_amd.fusion import MformerFusion
    args = types.SimpleNamespace(hidden_size=128, num_attention_heads=2, num_hidden_layers=1, use_intermediate=1, intermediate_size=400)
    assert set(MformerFusion(args, 4).state_dict().keys()) == set(params.keys())
    z = load("g1_fusion_m3_d128_h2.npz")
    _, params, cfg = fusion_inputs(z)
    args.use_intermediate = 0
    mod = MformerFusion(args, 3)
    assert set(mod.state_dict().keys()) == set(params.keys())
    assert sum(p.numel() for p in mod.parameters()) == 117894          # SURVEY.md par. 8b


def test_fusion_full_size_properties():
    """Config-2 shape (N=15000, M=3, D=128, H=2): size-independent properties of the operator."""
    from snag_amd.fusion import MformerFusion
    torch.manual_seed(0)
    args = types.SimpleNamespace(hidden_size=128, num_attention_heads=2, num_hidden_layers=1, use_intermediate=0, intermediate_size=400)
    mod = MformerFusion(args, 3).to(DEV).eval()
    N = 15000
    embs = [torch.randn(N, 128, device=DEV) * s for s in (1.0, 0.1, 3.0)]
    joint, joint_fz, hidden, w = mod(embs)
    assert_close(w.sum(1), torch.ones(N), atol=1e-5)                                   # softmax rows
    # each modality block of joint has norm w_m ; of joint_fz has norm softmax(weight_raw)_m = 1/6 at init
    nj = joint.view(N, 3, 128).norm(dim=2)
    assert_close(nj, w, atol=1e-5)
    assert_close(joint_fz.view(N, 3, 128).norm(dim=2), torch.full((N, 3), 1.0 / 6), atol=1e-5)
    # LayerNorm output with default affine: zero mean, unit (biased) variance per token
    assert float(hidden.mean(-1).abs().max()) < 1e-4
    assert_close(hidden.var(-1, unbiased=False), torch.ones(N, 3), atol=1e-3)
    # entity independence: a permutation of the entities permutes every output
    perm = torch.randperm(N, device=DEV)
    outs2 = mod([e[perm] for e in embs])
    for a, b in zip((joint, joint_fz, hidden, w), outs2):
        assert torch.equal(a[perm], b)
    # read-outs
    mod.joint_way = "Mformer_hd_graph"
    assert torch.equal(mod.readout(joint, joint_fz, hidden, graph_index=2), hidden[:, 2])
    mod.joint_way = "Mformer_hd_mean"
    assert_close(mod.readout(joint, joint_fz, hidden), hidden.mean(1), atol=1e-6)


def test_fusion_rejects_bad_input():
    from snag_amd.fusion import MformerFusion
    args = types.SimpleNamespace(hidden_size=128, num_attention_heads=3, num_hidden_layers=1, use_intermediate=0, intermediate_size=400)
    with pytest.raises(AssertionError):
        MformerFusion(args, 3)
    args.num_attention_heads = 2
    mod = MformerFusion(args, 3).to(DEV)
    with pytest.raises(ValueError):
        mod([None, None])
    with pytest.raises(ValueError):
        mod([torch.zeros(4, 64, device=DEV)])


def test_fusion_at_config5_scale_has_no_index_overflow():
    """BASELINE config 5 shape for the fusion: 1,000,000 entities x 3 tokens x 256 (2.3e9 qkv elements, past 2^31).  Slices of the
    full result must equal the same module run on just those entities, forward and backward."""
    import types
    from snag_amd.fusion import MformerFusion
    N, D = 1_000_000, 256
    a = types.SimpleNamespace(hidden_size=D, num_attention_heads=2, num_hidden_layers=1, use_intermediate=0, intermediate_size=400)
    m = MformerFusion(a, modal_num=3).to(DEV).eval()
    g = torch.Generator().manual_seed(0)
    embs = [torch.randn(N, D, generator=g).to(DEV).requires_grad_(True) for _ in range(3)]
    joint, joint_fz, hidden, w = m(embs + [None] * 3)
    (hidden.sum() + joint.sum()).backward()
    for lo in (0, 499_000, N - 1000):
        sub = [e[lo:lo + 1000].detach().clone().requires_grad_(True) for e in embs]
        j2, jf2, h2, w2 = m(sub + [None] * 3)
        (h2.sum() + j2.sum()).backward()
        for got, ref in ((h2, hidden[lo:lo + 1000]), (j2, joint[lo:lo + 1000]), (jf2, joint_fz[lo:lo + 1000]), (w2, w[lo:lo + 1000]),
                         (sub[0].grad, embs[0].grad[lo:lo + 1000]), (sub[2].grad, embs[2].grad[lo:lo + 1000])):
            assert float((got.detach() - ref.detach()).abs().max()) < 1e-6
    del embs, joint, joint_fz, hidden, w
    torch.cuda.empty_cache()


def test_prepare_reduce_reports_whether_the_weight_images_were_written():
    """snag_fusion_fwd_prepare_reduce = snag_gemm_reduce_group + the read-out forward's weight images in one launch.  It must say 0 -- and
    do nothing but the reduce -- for a descriptor that does not take the one-launch path (here: no co_fwd products), and the engine's
    step must not depend on who prepared the images: same read-out with the flag forced off (snag_fusion_fwd then prepares them itself)."""
    import ctypes
    from snag_amd import _lib
    from snag_amd.mkgc import MKGCConfig, MKGCHotPath
    N, R, B, K, D = 1100, 5, 64, 8, 128
    g = torch.Generator().manual_seed(5)
    # row count and feature widths at which both first projections share one grouped launch (its reduce is what the images ride in)
    feats = [torch.rand(N, 640, generator=g), torch.randn(N, 512, generator=g)]
    idx = [torch.randint(0, hi, sz, generator=g, dtype=torch.int32).to(DEV) for hi, sz in ((N, (B,)), (R, (B,)), (N, (B,)), (N, (B, K)))]
    outs = []
    for force_off in (False, True):
        cfg = MKGCConfig(n_ent=N, n_rel=R, dim=D, feat_dims=[640, 512], num_proj=2, joint_way="Mformer_hd_graph", margin=6.0, neg_num=K, dropout=0.1, seed=3)
        eng = MKGCHotPath(cfg, feats, device=DEV)
        eng.force_one_launch = True            # (1,100 entities: below the row threshold of the one-launch paths)
        if force_off:
            orig = eng._grouped_long_k
            eng._grouped_long_k = lambda *a, **kw: orig(*a, **{**kw, "prepare_fusion": False})
        loss = eng.forward(*idx, with_backward=True)
        torch.cuda.synchronize()
        outs.append((eng.entity_representation().clone(), loss.clone(), int(eng._fdesc.weight_images_ready), bool(eng.l1_grouped)))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][3] and outs[0][2] == 1 and outs[1][2] == 0          # prepared by the grouped launch's reduce / by snag_fusion_fwd itself
    # a descriptor without co_fwd products: nothing prepared, rc 0
    lib = _lib.load()
    fd = _lib.FusionDesc()
    ctypes.memmove(ctypes.byref(fd), ctypes.byref(eng._fdesc), ctypes.sizeof(fd))
    fd.n_co_fwd = 0
    ready = ctypes.c_int32(7)
    assert lib.snag_fusion_fwd_prepare_reduce(ctypes.byref(fd), None, 0, ctypes.byref(ready), None) == 0 and ready.value == 0


def test_strided_mask_region_draws_the_dense_regions_bytes():
    """snag_mask_region.period / skip / take: the read-out-only layer draws the output-dropout mask of ONE token (D bytes of every M D).
    Those bytes must be the dense region's (same Philox counters), everything else must be left alone, and a region that starts at a
    later global row must draw what the whole stream holds there (N ranks = 1 rank)."""
    import ctypes
    from snag_amd import _lib
    lib = _lib.load()
    rows, M, D, g = 37, 3, 128, 2
    n = rows * M * D

    def draw(offset, n_el, strided, fill):
        out = torch.full((n_el,), fill, dtype=torch.uint8, device=DEV)
        r = (_lib.MaskRegion * 1)()
        r[0].seed, r[0].offset, r[0].n, r[0].p, r[0].invert, r[0].out = 1234, offset, n_el, 0.3, 0, out.data_ptr()
        if strided:
            r[0].period, r[0].skip, r[0].take = M * D, g * D, D
        assert lib.snag_bernoulli_regions(r, 1, None, None) == 0, _lib.last_error()
        torch.cuda.synchronize()
        return out.view(-1, M, D)
    dense = draw(0, n, False, 7)
    strided = draw(0, n, True, 7)
    assert torch.equal(strided[:, g], dense[:, g]) and bool((strided[:, :g] == 7).all())
    assert 0.6 < float(dense.float().mean()) < 0.8
    lo = 11                                                   # a rank that owns rows 11 .. 36
    part = draw(lo * M * D, (rows - lo) * M * D, True, 9)
    assert torch.equal(part[:, g], dense[lo:, g]) and bool((part[:, :g] == 9).all())
    bad = (_lib.MaskRegion * 1)()
    bad[0].seed, bad[0].offset, bad[0].n, bad[0].p, bad[0].out = 1, 0, n, 0.3, dense.data_ptr()
    bad[0].period, bad[0].skip, bad[0].take = M * D, g * D + 2, D          # not multiples of 4 / past the period
    assert lib.snag_bernoulli_regions(bad, 1, None, None) != 0
