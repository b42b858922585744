"""GPU parity over SEQUENCES of engine calls: the single-step tests start from a freshly built engine, so state that one call leaves for
the next (the scorer's scatter targets, "already cleared" / "already prepared" flags, workspaces keyed on the batch size, the noise
generation, the dropout tick, train / eval mode) is never exercised by them.  Here a seeded random configuration -- every README
``joint_way``, both training scorers, both losses, one or two projection layers, FFN, one or two Mformer layers, dropout, noise -- runs a
seeded random sequence of

    train       forward(with_backward) + backward on a NEW batch of a new size     -> loss, scores, every gradient vs the oracle
    split       forward, then the two-launch backward                             -> the same
    abandoned   forward(with_backward=True) that no backward follows              -> (the next step must not see its leftovers)
    eval        eval-mode forward                                                 -> scores vs the oracle without noise / dropout
    optimizer   clip + AdamW (the oracle re-reads the engine's parameters)
    noise       update_noise()                                                    -> the next step runs on the new noisy tables
    linkpred    filtered link prediction                                          -> ranks vs the oracle's count-less-than

and every checked call is compared with the CPU oracle evaluated at the engine's CURRENT parameters, noisy tables and keep masks
(reference: model/SNAG_tools.py:23-51,122-152, model/MSNEA.py:110-127 -- through oracle/ref_cpu.py).  ``tools/soak_engine_sequences.py``
runs ``_run`` on seeds the suite does not use."""
import random

import pytest
import torch

from helpers import assert_close
from oracle import ref_cpu as O
from test_mkgc_gpu import engine_masks, oracle_params

pytestmark = pytest.mark.gpu
DEV = "cuda"

JOINT = ["Mformer_hd_graph", "Mformer_hd_mean", "Mformer_weight", "learnable_weight", "atten_weight"]


def _config(rnd):
    from snag_amd.mkgc import MKGCConfig
    joint_way = rnd.choice(JOINT)
    D = rnd.choice([64, 128, 128])
    N = rnd.choice([rnd.randint(40, 90), rnd.randint(200, 500)])
    layers = 1 if joint_way == "atten_weight" else rnd.choice([1, 1, 2])
    ffn = 0 if joint_way == "atten_weight" else rnd.choice([0, 0, 1])
    loss = rnd.choice(["hinge", "hinge", "self_adversarial"])
    scorer = rnd.choice(["segmented", "atomic"])
    dropout = rnd.choice([0.0, 0.1])
    num_proj = rnd.choice([1, 2])
    feat_dims = [rnd.choice([24, 40, 72]), rnd.choice([12, 40])]
    if rnd.random() < 0.3:          # wide tables over >= 512 entities: the first projections (and their weight gradients) go out as the grouped launch of
        N, feat_dims = rnd.randint(520, 700), [rnd.choice([1030, 1100]), rnd.choice([520, 600])]          # eight-wave tiles -- with the backward's reduce riding in it
    return MKGCConfig(n_ent=N, n_rel=rnd.randint(1, 9), dim=D, feat_dims=feat_dims, num_proj=num_proj,
                      joint_way=joint_way, num_heads=rnd.choice([1, 2, 4]), num_layers=layers, use_intermediate=ffn, intermediate_size=96,
                      margin=6.0, loss=loss, adv_temp=0.7, neg_num=rnd.choice([1, 8, 32]), add_noise=rnd.choice([0, 1]), dropout=dropout,
                      proj_dropout=rnd.choice([0.0, 0.2]) if (num_proj == 2 and dropout > 0) else 0.0, lr=2e-3, weight_decay=0.05, clip=1.0,
                      readout_only=rnd.random() < 0.8, fold_proj=rnd.random() < 0.8, scorer=scorer,
                      deterministic=(scorer == "atomic" and rnd.random() < 0.3), seed=rnd.randint(1, 1 << 20))


def _batch(cfg, g, B):
    N, R, K = cfg.n_ent, cfg.n_rel, cfg.neg_num
    return (torch.randint(0, N, (B,), generator=g, dtype=torch.int32), torch.randint(0, R, (B,), generator=g, dtype=torch.int32),
            torch.randint(0, N, (B,), generator=g, dtype=torch.int32), torch.randint(0, N, (B, K), generator=g, dtype=torch.int32))


def _oracle(eng, batch, train):
    """The oracle's step at the engine's current parameters -- in training mode on its noisy tables and with the keep masks it drew last."""
    cfg = eng.cfg
    h, r, t, neg = batch
    p, proj, fp = oracle_params(eng)
    noisy = bool(train and cfg.add_noise and eng.noise_active)
    drop = train and eng.n_layers > 0 and (cfg.dropout > 0 or cfg.proj_dropout > 0)
    masks, proj_keep = engine_masks(eng) if drop else (None, None)
    if proj_keep is None and train and eng.keep_proj is not None:          # atten_weight keeps the ProjectionHead dropout and has no Mformer masks
        proj_keep = [eng.keep_proj[i][:eng.shard.n_loc].cpu().float() for i in range(len(cfg.feat_dims))]
    if cfg.dropout == 0:
        masks = None
    feats = [f.cpu() for f in (eng.noisy if noisy else eng.feats)]
    out = O.mkgc_forward(feats, proj, p["ent"], fp, p["rel"], h, r, t, neg, num_heads=cfg.num_heads, margin=cfg.margin, joint_way=cfg.joint_way,
                         use_intermediate=bool(cfg.use_intermediate), num_layers=cfg.num_layers,
                         ent_noise=eng.ent_noise.cpu() if noisy else None, ent_noise_mask=eng.ent_noise_mask.cpu().bool() if noisy else None,
                         mask_ratio=cfg.mask_ratio, loss_kind=cfg.loss, adv_temp=cfg.adv_temp, drop_masks=masks, drop_p=cfg.dropout,
                         proj_keep=proj_keep, proj_p=cfg.proj_dropout, atten_a=p.get("atten.a"))
    return p, out


def _check_train(eng, batch, fused, what):
    loss, pos, ng = eng.forward(*[x.to(DEV) for x in batch], want_scores=True, with_backward=fused)
    eng.backward()
    torch.cuda.synchronize()
    p, (oloss, opos, oneg, _) = _oracle(eng, batch, True)
    oloss.backward()
    assert_close(pos, opos, atol=1e-4, what=f"{what}: pos")                  # north_star: scores within 1e-4
    assert_close(ng, oneg, atol=1e-4, what=f"{what}: neg")
    assert_close(loss, oloss, atol=2e-4, rtol=4e-6, what=f"{what}: loss")
    grads = eng.named_gradients()
    cfg = eng.cfg
    for k, ref in p.items():
        gref = ref.grad if ref.grad is not None else torch.zeros_like(ref)
        got = grads[k].cpu()
        if cfg.num_proj == 2 and k.endswith(".l1"):
            # ReLU kink: a first-projection pre-activation within rounding of zero (|z| < 1e-6; seen at 1e-8 in 2 of 600 soak sequences) is
            # positive on one side and zero on the other, and the whole row of that unit's weight gradient gains or loses one entity's term.
            # Such units are left out of the comparison (nothing else depends on the sign: mid itself is ~0 there).
            i = int(k[4:k.index(".")])
            noisy = bool(cfg.add_noise and eng.noise_active)
            z = (eng.noisy if noisy else eng.feats)[i].cpu().double() @ ref.detach().double().T
            sure = z.abs().min(dim=0).values > 1e-6 * max(1.0, float(z.abs().max()))
            got, gref = got[sure], gref[sure]
        assert_close(got, gref, atol=2e-4 * max(1.0, float(gref.abs().max())) if gref.numel() else 2e-4, rtol=1e-4, what=f"{what}: grad {k}")


def _run(seed, n_ops=8, verbose=False):
    from snag_amd.mkgc import MKGCHotPath
    rnd = random.Random(seed)
    cfg = _config(rnd)
    g = torch.Generator().manual_seed(seed)
    feats = [torch.rand(cfg.n_ent, cfg.feat_dims[0], generator=g), torch.randn(cfg.n_ent, cfg.feat_dims[1], generator=g)]
    eng = MKGCHotPath(cfg, feats, device=DEV)
    eng.p["ent"].copy_(torch.randn(cfg.n_ent, cfg.dim, generator=g).to(DEV))
    eng.p["weight_raw"].copy_(torch.randn(6, generator=g).to(DEV))
    if cfg.joint_way == "atten_weight":
        eng.p["atten.a"].copy_((torch.randn(cfg.dim, generator=g) * (2.0 / cfg.dim ** 0.5)).to(DEV))
    fused_ok = cfg.loss == "hinge"                 # (the self-adversarial loss has no one-call training form: forward, then backward)
    ops = ["train"] + [rnd.choice(["train", "train", "split", "abandoned", "eval", "optimizer", "noise", "linkpred"]) for _ in range(n_ops - 2)] + ["train"]
    trace = []
    for i, op in enumerate(ops):
        B = rnd.choice([1, 7, 33, 96, 130])
        what = f"seed {seed} op {i} {op} (B={B}; {cfg.joint_way}, {cfg.scorer}{'/det' if cfg.deterministic else ''}, {cfg.loss}, proj {cfg.num_proj}, L {cfg.num_layers}, " \
               f"ffn {cfg.use_intermediate}, ro {int(cfg.readout_only)}, fold {int(cfg.fold_proj)}, drop {cfg.dropout}/{cfg.proj_dropout}, noise {cfg.add_noise}, N {cfg.n_ent}, " \
               f"D {cfg.dim}, H {cfg.num_heads}, K {cfg.neg_num}; after {trace})"
        if verbose:
            print(what)
        if op == "train":
            _check_train(eng, _batch(cfg, g, B), fused_ok, what)
        elif op == "split":
            _check_train(eng, _batch(cfg, g, B), False, what)
        elif op == "abandoned":
            eng.forward(*[x.to(DEV) for x in _batch(cfg, g, B)], with_backward=fused_ok)
        elif op == "eval":
            batch = _batch(cfg, g, B)
            eng.eval()
            loss, pos, ng = eng.forward(*[x.to(DEV) for x in batch], want_scores=True)
            eng.train()
            with torch.no_grad():
                _, (oloss, opos, oneg, _) = _oracle(eng, batch, False)
            assert_close(pos, opos, atol=1e-4, what=f"{what}: pos")
            assert_close(ng, oneg, atol=1e-4, what=f"{what}: neg")
            assert_close(loss, oloss, atol=2e-4, rtol=4e-6, what=f"{what}: loss")
        elif op == "optimizer":
            if trace and trace[-1] in ("train", "split"):          # (gradients of a complete backward are in place)
                eng.optimizer_step()
            else:
                op = "optimizer(skipped)"
        elif op == "noise":
            if cfg.add_noise:
                eng.update_noise()
        elif op == "linkpred":
            h, r, t, _ = _batch(cfg, g, B)
            res = eng.link_prediction(h.to(DEV), r.to(DEV), t.to(DEV))
            with torch.no_grad():
                p, (_, _, _, oent) = _oracle(eng, (h, r, t, torch.zeros(B, cfg.neg_num, dtype=torch.int32)), False)
                oref = O.rank_full_table(oent.double(), p["rel"].double(), h, r, t)
            got, sc = res["rank"].cpu().long(), res["scores"].cpu()
            near = ((sc - sc.gather(1, t.long().view(-1, 1))).abs() < 1e-5).sum(1) - 1          # candidates within 1e-5 of the true tail's score
            assert bool(((got - oref).abs() <= near).all()), f"{what}: rank differs beyond near-ties by {int(((got - oref).abs() - near).max())}"
        trace.append(op)
    return cfg, trace


@pytest.mark.parametrize("seed", list(range(100, 124)) + [1318, 1473])
def test_engine_call_sequences_vs_oracle(seed):
    _run(seed)
