"""CPU: pin oracle/ref_cpu.py against the golden vectors produced by the imported reference.

The reference has no tests of its own (SURVEY.md par. 4); these fixtures are the pins (par. 8c).
"""
import numpy as np
import pytest
import torch

from helpers import ATOL, T, assert_close, fusion_cases, fusion_inputs, fusion_masks, load
from oracle import ref_cpu as O

torch.set_num_threads(4)
OUTS = ("joint", "joint_fz", "hidden", "weight_norm")


@pytest.mark.parametrize("case", fusion_cases())
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_fusion_forward_backward(case, mode):
    z = load(case)
    embs, params, cfg = fusion_inputs(z, requires_grad=True)
    masks = fusion_masks(z, cfg["L"]) if mode == "train" else None
    outs = O.fusion_forward(embs, params, cfg["H"], cfg["L"], cfg["ffn"], drop_masks=masks)
    for nm, o in zip(OUTS, outs):
        assert_close(o, z[f"{mode}.{nm}"], atol=2e-6, rtol=2e-6, what=f"{case}/{mode}/{nm}")
    scalar = sum((o * T(z[f"cw{i}"])).sum() for i, o in enumerate(outs))
    scalar.backward()
    for i, e in enumerate(embs):
        if e is not None:
            assert_close(e.grad, z[f"{mode}.g.emb{i}"], atol=2e-5, rtol=1e-4, what=f"{case}/{mode}/g.emb{i}")
    for k, p in params.items():
        key = f"{mode}.g.p.{k}"
        if key in z:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            assert_close(g, z[key], atol=5e-5, rtol=1e-4, what=key)


def test_readouts():
    z = load("g1_fusion_m3_d128_h2.npz")
    hid, j, jf = T(z["eval.hidden"]), T(z["eval.joint"]), T(z["eval.joint_fz"])
    assert torch.equal(O.readout("Mformer_hd_graph", j, jf, hid, 2), hid[:, 2])
    assert_close(O.readout("Mformer_hd_mean", j, jf, hid, 2), hid.mean(1), atol=0)
    assert O.readout("Mformer_weight", j, jf, hid, 2) is j
    assert O.readout("learnable_weight", j, jf, hid, 2) is jf
    with pytest.raises(ValueError):
        O.readout("nope", j, jf, hid, 0)


def test_noise():
    z = load("g2_noise.npz")
    eps = float(z["mask_ratio"])
    m, s = O.col_mean_std(T(z["img"]), T(z["ent_wo_img"]))
    assert_close(m, z["img_mean"], atol=1e-7)
    assert_close(s, z["img_std"], atol=1e-7)
    for nm in ("rel", "att"):
        m, s = O.col_mean_std(T(z[nm]))
        assert_close(m, z[f"{nm}_mean"], atol=1e-6)
        assert_close(s, z[f"{nm}_std"], atol=1e-6)
    for nm in ("rel", "att", "img"):
        out = O.add_noise_rows(T(z[nm]), T(z[f"{nm}_mean"]), T(z[f"{nm}_std"]), T(z[f"{nm}_mask"]), T(z[f"{nm}_z"]), eps)
        assert torch.equal(out, T(z[f"{nm}_noisy"])), nm     # bit-exact replay of the reference RNG stream
    m, s = O.col_mean_std(T(z["ent_w"]))
    assert_close(m, z["ent_mean"], atol=1e-7)
    assert_close(s, z["ent_std"], atol=1e-7)
    assert torch.equal(T(z["ent_mean"]) + T(z["ent_std"]) * T(z["ent_z"]), T(z["entity_noise"]))
    assert np.array_equal(z["ent_u"] < float(z["noise_ratio"]) * 0.5, z["entity_noise_mask"])


def test_score_intree():
    z = load("g3_score.npz")
    e = T(z["ent"]).requires_grad_(True)
    r = T(z["rel"]).requires_grad_(True)
    idx = [T(z[k]) for k in ("p_h", "p_r", "p_t", "n_h", "n_r", "n_t")]
    loss, pos, neg = O.kge_intree(e, r, *idx, float(z["margin"]))
    assert_close(pos, z["pos"], atol=1e-6)
    assert_close(neg, z["neg"], atol=1e-6)
    assert_close(loss, z["loss_term1"], atol=1e-4)
    loss.backward()
    assert_close(e.grad, z["g.ent"], atol=1e-5)
    assert_close(r.grad, z["g.rel"], atol=1e-5)
    # K=1 negative-tail form reduces to the in-tree pairing when only tails are corrupted
    tails_only = torch.from_numpy(z["n_h"] == z["p_h"])
    sel = tails_only.nonzero().flatten()
    l2, p2, n2 = O.score_margin(e.detach(), r.detach(), idx[0][sel], idx[1][sel], idx[2][sel], idx[5][sel].view(-1, 1),
                                float(z["margin"]))
    assert_close(p2, z["pos"][sel.numpy()], atol=1e-6)
    assert_close(n2.flatten(), z["neg"][sel.numpy()], atol=1e-6)


@pytest.mark.parametrize("name", ["d64", "d1200", "d64_nocsls"])
def test_eval(name):
    z = load(f"g4_eval_{name}.npz")
    final = O.nrm(T(z["emb"]))
    left, right = T(z["left"]), T(z["right"])
    k = int(z["csls_k"])
    assert_close(O.pairwise_distances(final[left], final[right]), z["dist"], atol=1e-6)
    if k:
        assert_close(O.csls_sim(1 - T(z["dist"]), k), z["csls"], atol=1e-6)
    res = O.rank_metrics(final[left], final[right], csls_k=k)
    assert np.array_equal(res["rank_l2r"].numpy(), z["rank_l2r"])           # ranking indices: bit-exact, both directions
    assert np.array_equal(res["rank_r2l"].numpy(), z["rank_r2l"])           # (r2l = the reference's own per-column sorts, main.py:422-424)
    for d in ("l2r", "r2l"):
        assert np.array_equal(np.asarray(res[f"hits_{d}"]), z[f"hits_{d}"]), (res[f"hits_{d}"], z[f"hits_{d}"])
        assert res[f"mr_{d}"] == pytest.approx(float(z[f"mr_{d}"]), rel=1e-13)         # unrounded; only the fp64 summation order differs
        assert res[f"mrr_{d}"] == pytest.approx(float(z[f"mrr_{d}"]), rel=1e-13)
        assert f"{res[f'mr_{d}']:.3f}" == f"{float(z[f'mr_{d}_3dp']):.3f}" and f"{res[f'mrr_{d}']:.3f}" == f"{float(z[f'mrr_{d}_3dp']):.3f}"
    assert 0.05 < res["hits_l2r"][0] < 0.999        # the fixture is informative


def test_mutual_nearest():
    z = load("g5_il.npz")
    _, _, pairs = O.mutual_nearest(T(z["final"]), z["left"].tolist(), z["right"].tolist())
    assert np.array_equal(np.array(pairs, dtype=np.int64), z["links"])
    assert len(pairs) > 50


def test_projections():
    z = load("g6_proj.npz")
    x = T(z["x"]).requires_grad_(True)
    w, b = T(z["lin.weight"]).requires_grad_(True), T(z["lin.bias"]).requires_grad_(True)
    y = O.linear(x, w, b)
    assert_close(y, z["lin.y"], atol=2e-6)
    (y * T(z["cw"])).sum().backward()
    assert_close(w.grad, z["lin.g.weight"], atol=1e-5)
    assert_close(b.grad, z["lin.g.bias"], atol=1e-5)
    for mode in ("eval", "train"):
        x = T(z["x"]).requires_grad_(True)
        w1, w2 = T(z["ph.l1"]).requires_grad_(True), T(z["ph.l2"]).requires_grad_(True)
        mask = T(z["ph.mask"]) if mode == "train" else None
        y = O.projection_head(x, w1, w2, mask, float(z["ph.p"]))
        assert_close(y, z[f"ph.{mode}.y"], atol=2e-6)
        (y * T(z["cw"])).sum().backward()
        assert_close(w1.grad, z[f"ph.{mode}.g.l1"], atol=1e-5)
        assert_close(w2.grad, z[f"ph.{mode}.g.l2"], atol=1e-5)
        assert_close(x.grad, z[f"ph.{mode}.g.x"], atol=1e-5)


@pytest.mark.parametrize("tag", ["clean", "noisy"])
def test_encoder(tag):
    """MultiModalEncoder.forward (SNAG_tools.py:122-156): projections + entity noise blend + fusion."""
    z = load("g7_encoder.npz")
    P = {k[2:]: T(v) for k, v in z.items() if k.startswith("p.")}
    ent = P["entity_emb.weight"].clone().requires_grad_(True)
    e = ent
    if tag == "noisy":
        e = O.entity_noise_blend(ent, T(z["entity_noise"]), T(z["entity_noise_mask"]), float(z["mask_ratio"]))
    img = O.linear(T(z["img"]), P["img_fc.weight"], P["img_fc.bias"])
    rel = O.linear(T(z["rel"]), P["rel_fc.weight"], P["rel_fc.bias"])
    att = O.linear(T(z["att"]), P["att_fc.weight"], P["att_fc.bias"])
    for nm, v in (("img", img), ("rel", rel), ("att", att), ("gph", e)):
        assert_close(v, z[f"{tag}.{nm}"], atol=2e-6, what=nm)
    fp = {k[len("fusion."):]: v for k, v in P.items() if k.startswith("fusion.")}
    outs = O.fusion_forward([img, att, rel, e, None, None], fp, num_heads=2)
    for nm, o in zip(OUTS, outs):
        assert_close(o, z[f"{tag}.{nm}"], atol=3e-6, what=nm)
    outs[2][:, 3, :].sum().backward()
    assert_close(ent.grad, z[f"{tag}.g.entity_emb"], atol=1e-5)


@pytest.mark.parametrize("tag", ["plain", "weighted", "asym"])
def test_icl_loss(tag):
    z = load("g8_icl_loss.npz")
    emb = T(z["emb"]).requires_grad_(True)
    w = T(z["weight_norm"]) if tag == "weighted" else None
    loss = O.icl_loss(emb, T(z["links"]), float(z[f"{tag}.tau"]), float(z[f"{tag}.ab"]), weight_norm=w)
    assert_close(loss, z[f"{tag}.loss"], atol=2e-6, rtol=2e-6)
    loss.backward()
    assert_close(emb.grad, z[f"{tag}.g.emb"], atol=2e-6, rtol=1e-4)


def test_multi_loss_layer():
    z = load("g8_icl_loss.npz")
    lv = T(z["ml.log_vars"]).requires_grad_(True)
    ls = [l.clone().requires_grad_(True) for l in T(z["ml.losses"])]
    tot = O.multi_loss(ls, lv)
    assert_close(tot, z["ml.total"], atol=1e-6)
    tot.backward()
    assert_close(lv.grad, z["ml.g.log_vars"], atol=1e-6)
    assert_close(torch.stack([l.grad for l in ls]), z["ml.g.losses"], atol=1e-6)


def test_gat():
    z = load("g9_gat.npz")
    x = T(z["x"]).requires_grad_(True)
    layers = [{"w": T(z[f"p.layer_stack.{l}.w"]).requires_grad_(True), "a_src_dst": T(z[f"p.layer_stack.{l}.a_src_dst"]).requires_grad_(True)}
              for l in range(2)]
    y = O.gat_forward(x, T(z["edges"]), layers)
    assert_close(y, z["y"], atol=2e-6, rtol=1e-5)
    (y * T(z["cw"])).sum().backward()
    assert_close(x.grad, z["g.x"], atol=2e-6, rtol=1e-4)
    for l in range(2):
        assert_close(layers[l]["w"].grad, z[f"g.p.layer_stack.{l}.w"], atol=2e-5, rtol=1e-4)
        assert_close(layers[l]["a_src_dst"].grad, z[f"g.p.layer_stack.{l}.a_src_dst"], atol=2e-5, rtol=1e-4)


def test_gcn():
    """Golden G12: the reference GCN (structure_encoder == "gcn") forward and autograd."""
    z = load("g12_gcn.npz")
    x = T(z["x"]).requires_grad_(True)
    adj = torch.sparse_coo_tensor(T(z["adj_indices"]), T(z["adj_values"]), (x.shape[0], x.shape[0]))
    p = {k[2:]: T(v).requires_grad_(True) for k, v in z.items() if k.startswith("p.")}
    y = O.gcn_forward(x, adj, p)
    assert_close(y, z["y"], atol=1e-6, rtol=1e-5)
    (y * T(z["cw"])).sum().backward()
    assert_close(x.grad, z["g.x"], atol=1e-6, rtol=1e-5)
    for k, v in p.items():
        assert_close(v.grad, z["g.p." + k], atol=1e-5, rtol=1e-5, what=k)


def test_mkgc_forward_passes_the_keep_masks_through():
    """oracle.mkgc_forward(drop_masks=, proj_keep=) -- what the GPU dropout-parity tests feed the engine's masks into -- is the composition of the
    pinned pieces: projection_head with its keep mask (G6), fusion_forward with the Bert masks (G1-train), the read-out, score_margin (G3)."""
    import torch
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(3)
    N, D, H, M, B, K = 40, 16, 2, 3, 8, 4
    feats = [torch.rand(N, 10, generator=g), torch.randn(N, 6, generator=g)]
    proj = [{"l1": torch.randn(D, 10, generator=g), "l2": torch.randn(D, D, generator=g)}, {"l1": torch.randn(D, 6, generator=g), "l2": torch.randn(D, D, generator=g)}]
    fp = {"weight_raw": torch.randn(6, generator=g)}
    pre = "fusion_layer.0."
    for nm in ("query", "key", "value"):
        fp[pre + f"attention.self.{nm}.weight"], fp[pre + f"attention.self.{nm}.bias"] = torch.randn(D, D, generator=g) * 0.3, torch.randn(D, generator=g) * 0.1
    fp[pre + "attention.output.dense.weight"], fp[pre + "attention.output.dense.bias"] = torch.randn(D, D, generator=g) * 0.3, torch.randn(D, generator=g) * 0.1
    fp[pre + "attention.output.LayerNorm.weight"], fp[pre + "attention.output.LayerNorm.bias"] = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    ent, rel = torch.randn(N, D, generator=g), torch.randn(3, D, generator=g)
    masks = [{"probs": (torch.rand(N, H, M, M, generator=g) > 0.1).float(), "attn_out": (torch.rand(N, M, D, generator=g) > 0.1).float()}]
    keep = [(torch.rand(N, D, generator=g) > 0.2).float() for _ in range(2)]
    h, r, t = torch.randint(0, N, (B,), generator=g), torch.randint(0, 3, (B,), generator=g), torch.randint(0, N, (B,), generator=g)
    neg = torch.randint(0, N, (B, K), generator=g)
    a = torch.randn(D, generator=g) * 0.3
    for jw in O.JOINT_WAYS:
        rl = rel if (jw.startswith("Mformer_hd") or jw == "atten_weight") else torch.randn(3, M * D, generator=g)
        loss, pos, ng, rep = O.mkgc_forward(feats, proj, ent, fp, rl, h, r, t, neg, num_heads=H, margin=4.0, joint_way=jw, drop_masks=masks, drop_p=0.1,
                                            proj_keep=keep, proj_p=0.2, atten_a=a)
        toks = [O.projection_head(f, p["l1"], p["l2"], keep_mask=k, p=0.2) for f, p, k in zip(feats, proj, keep)] + [ent]
        joint, joint_fz, hidden, _ = O.fusion_forward(toks, fp, H, 1, False, drop_masks=masks, drop_p=0.1)
        want = O.atten_weight_readout(torch.stack(toks, 1), a) if jw == "atten_weight" else O.readout(jw, joint, joint_fz, hidden, graph_index=M - 1)
        l2, p2, n2 = O.score_margin(want, rl, h, r, t, neg, 4.0)
        assert torch.equal(rep, want) and torch.equal(loss, l2) and torch.equal(pos, p2) and torch.equal(ng, n2)
        # and the masks matter: without them the read-out differs
        assert not torch.equal(O.mkgc_forward(feats, proj, ent, fp, rl, h, r, t, neg, num_heads=H, margin=4.0, joint_way=jw, atten_a=a)[3], want)


@pytest.mark.parametrize("tag", ["m3_d128", "m4_d300", "m6_d64"])
def test_atten_weight_readout(tag):
    """G15: joint_way "atten_weight" -- the oracle's restatement against EVA_tools.SelfAttention.forward run from the reference tree
    (model/EVA_tools.py:190-232): representation and, through autograd, both gradients."""
    z = load("g15_atten_weight.npz")
    x, a = T(z[f"{tag}.x"]).requires_grad_(True), T(z[f"{tag}.a"]).requires_grad_(True)
    rep = O.atten_weight_readout(x, a)
    assert_close(rep, z[f"{tag}.rep"], atol=1e-5, rtol=1e-5, what="rep")          # (pre-activations of size ~10 go through exp: fp32 noise of the dot product x e^s)
    (rep * T(z[f"{tag}.cw"])).sum().backward()
    assert_close(x.grad, z[f"{tag}.g.x"], atol=2e-5, rtol=1e-4, what="g.x")
    assert_close(a.grad, z[f"{tag}.g.a"], atol=2e-4, rtol=1e-4, what="g.a")


def test_adaptive_avg_pool_rows_windows():
    """README POOL=1: the oracle's pool is torch's adaptive windows; restated here by hand for the 1-D case and a 2-D view."""
    g = torch.Generator().manual_seed(2)
    x = torch.randn(5, 770, generator=g)
    P = 256
    got = O.adaptive_avg_pool_rows(x, P)
    for i in (0, 1, 100, 255):
        lo, hi = (i * 770) // P, -(-((i + 1) * 770) // P)
        assert_close(got[:, i], x[:, lo:hi].mean(dim=1), atol=1e-6, what=f"window {i}")
    y = torch.randn(3, 12 * 64, generator=g)
    got2 = O.adaptive_avg_pool_rows(y, (4, 64), view_hw=(12, 64))
    assert_close(got2, y.view(3, 4, 3, 64).mean(dim=2).reshape(3, 256), atol=1e-6, what="12x64 -> 4x64")


def test_gat_with_dropout():
    """G16: the reference GAT in TRAIN mode (dropout 0.2 on the first layer's input, attn_dropout 0.3 on every head's edge weights after the row
    sums; F.dropout replaced by recorded masks): oracle.gat_forward with the same masks, output and all gradients."""
    z = load("g16_gat_dropout.npz")
    x = T(z["x"]).requires_grad_(True)
    layers = [{"w": T(z[f"p.layer_stack.{l}.w"]).requires_grad_(True), "a_src_dst": T(z[f"p.layer_stack.{l}.a_src_dst"]).requires_grad_(True)}
              for l in range(2)]
    y = O.gat_forward(x, T(z["edges"]), layers, in_keep=[T(z["in_keep0"]), None], edge_keep=[T(z["edge_keep0"]), T(z["edge_keep1"])],
                      p_in=float(z["p_in"]), p_edge=float(z["p_edge"]))
    assert_close(y, z["y"], atol=2e-6, rtol=1e-5)
    (y * T(z["cw"])).sum().backward()
    assert_close(x.grad, z["g.x"], atol=2e-6, rtol=1e-4)
    for l in range(2):
        assert_close(layers[l]["w"].grad, z[f"g.p.layer_stack.{l}.w"], atol=2e-5, rtol=1e-4)
        assert_close(layers[l]["a_src_dst"].grad, z[f"g.p.layer_stack.{l}.a_src_dst"], atol=2e-5, rtol=1e-4)
    # the masks matter
    assert not torch.allclose(O.gat_forward(x.detach(), T(z["edges"]), [{k: v.detach() for k, v in l.items()} for l in layers]), T(z["y"]), atol=1e-3)
