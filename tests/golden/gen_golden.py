#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  It executes the reference's own Python (CPU, fp32) on seeded
synthetic inputs and stores inputs + outputs as .npz.  Nothing from the
reference's source text is written anywhere: the fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Vectors (SURVEY.md par. 8c):
  g1_fusion_*.npz    MformerFusion.forward eval / train(dropout masks) + autograd grads
  g2_noise.npz       SNAG.add_noise_to_embeddings / get_mean_std / update_noise
  g3_score.npz       MSNEA.kge relation loss (term 1) + grads
  g4_eval_*.npz      pairwise_distances / csls_sim / Runner._test metrics + per-row (l2r) and per-column (r2l) ranks, MR / MRR unrounded
  g5_il.npz          SNAG.Iter_new_links mutual nearest neighbours
  g6_proj.npz        nn.Linear projections + ProjectionHead fwd/grads
  g7_encoder.npz     MultiModalEncoder.forward (projections + entity noise blend + fusion), GAT stubbed to identity
  g9_gat.npz         GAT(diag=True) structural encoder (model/layers.py:35-100, model/Tool_model.py:61-94), output and grads
  g10_snag_model.npz the whole in-tree SNAG model: SNAG.forward/update_noise/joint_emb_generat (model/SNAG.py) loss + grads
  g11_data.npz       dataset readers on a tiny synthetic DBP15K-layout directory: read_raw_data / load_relation / load_attr / load_img /
                     get_adjr / load_eva_data (src/data.py, src/utils.py:327-362); the fixture holds the synthetic files and the outputs
  g13_data_surface_unsup.npz  loader options: GloVe name / character-bigram features (--use_surface) and unsupervised visual pivots (--unsup)
  g14_set_optim.npz  set_optim parameter groups / cosine schedule on the SNAG model (src/utils.py:25-86)
  g12_gcn.npz        GCN structural encoder option (model/Tool_model.py:98-110, model/layers.py:102-130), output and grads
  g8_icl_loss.npz    icl_loss (bidirectional InfoNCE, model/SNAG_loss.py:31-128) + CustomMultiLossLayer, loss and grads
  g16_gat_dropout.npz   the same GAT in TRAIN mode with dropout 0.2 / attn_dropout 0.3 (F.dropout replaced by recorded masks): output, masks, grads
  g15_atten_weight.npz  EVA_tools.SelfAttention.forward (model/EVA_tools.py:190-232; the in-tree arithmetic of joint_way "atten_weight"), output, weights, grads
"""
import csv
import io
import logging
import os
import re
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference/SNAG_MMEA"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True

if not os.path.isdir(REF):
    sys.exit("reference tree not present; golden vectors can only be generated in the build container")

# --- SURVEY.md Appendix A import recipe ------------------------------------
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
for name, attrs in [("easydict", {"EasyDict": dict}),
                    ("unidecode", {"unidecode": lambda s: s}),
                    ("torch.utils.tensorboard", {"SummaryWriter": object})]:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
sys.path.insert(0, REF)
from model.SNAG_tools import MformerFusion, MultiModalEncoder  # noqa: E402
from model.SNAG import SNAG  # noqa: E402
from model.MSNEA import MSNEA  # noqa: E402
from model.layers import ProjectionHead  # noqa: E402
from src.utils import pairwise_distances, csls_sim, Loss_log  # noqa: E402
import main as ref_main  # noqa: E402
import torch.nn.functional as F  # noqa: E402

torch.set_num_threads(1)   # deterministic reduction order for the stored vectors


def npify(d):
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def save(name, d):
    path = os.path.join(OUT, name)
    np.savez(path, **npify(d))
    print(f"{name:32s} {os.path.getsize(path) / 1024:8.1f} KB")


def fusion_args(D, H, I, ffn, L=1):
    return types.SimpleNamespace(hidden_size=D, num_attention_heads=H, num_hidden_layers=L,
                                 use_intermediate=ffn, intermediate_size=I)


class DropoutRecorder:
    """Replace F.dropout by an explicit-mask inverted dropout and record the masks in call order."""

    def __init__(self, gen):
        self.gen = gen
        self.masks = []
        self._orig = F.dropout

    def __enter__(self):
        def fake(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            keep = (torch.rand(x.shape, generator=self.gen) >= p)
            self.masks.append(keep.to(torch.uint8))
            return x * keep.to(x.dtype) / (1.0 - p)
        F.dropout = fake
        torch.nn.functional.dropout = fake
        return self

    def __exit__(self, *a):
        F.dropout = self._orig
        torch.nn.functional.dropout = self._orig


def randomize_(module, gen, scale=1.0):
    """Non-trivial parameters everywhere (LayerNorm affine, biases, weight_raw included)."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            if "LayerNorm.weight" in n:
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            elif n.endswith("weight_raw"):
                p.copy_(torch.randn(p.shape, generator=gen))
            elif p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=gen) * (scale / np.sqrt(p.shape[-1])))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))


# --------------------------------------------------------------------------
def gen_g1():
    cases = [
        # name,              N,  M, D,   H, I,   ffn, L, pattern of None slots (6), modes that store parameter grads
        ("m3_d128_h2",       64, 3, 128, 2, 400, 0, 1, [1, 1, 1, 0, 0, 0], ("eval", "train")),
        ("m3_d128_h2_skip",  50, 3, 128, 2, 400, 0, 1, [1, 0, 1, 1, 0, 0], ()),
        ("m4_d128_h2_ffn",   48, 4, 128, 2, 400, 1, 1, [1, 1, 1, 1, 0, 0], ("train",)),
        ("m6_d128_h2_ffn",   40, 6, 128, 2, 400, 1, 1, [1, 1, 1, 1, 1, 1], ()),
        ("m4_d300_h1_ffn",   40, 4, 300, 1, 400, 1, 1, [1, 1, 1, 1, 0, 0], ("eval",)),
        ("m3_d64_h4_l2",     33, 3, 64,  4, 96,  1, 2, [1, 1, 1, 0, 0, 0], ("eval", "train")),
    ]
    for ci, (name, N, M, D, H, I, ffn, L, pat, pgrad_modes) in enumerate(cases):
        g = torch.Generator().manual_seed(1000 + ci)
        fus = MformerFusion(fusion_args(D, H, I, ffn, L), modal_num=M)
        randomize_(fus, g)
        scales = [1.0, 0.3, 2.0, 0.05, 1.5, 0.7]
        embs = [(torch.randn(N, D, generator=g) * scales[i]).requires_grad_(True) if pat[i] else None for i in range(6)]
        if name == "m3_d128_h2":           # one all-zero row exercises the normalize eps branch
            with torch.no_grad():
                embs[1][7].zero_()
        # fixed linear functional of the four outputs -> scalar for the gradient check
        cw = [torch.randn(N, M * D, generator=g), torch.randn(N, M * D, generator=g),
              torch.randn(N, M, D, generator=g), torch.randn(N, M, generator=g)]
        d = {"N": N, "M": M, "D": D, "H": H, "I": I, "ffn": ffn, "L": L, "pattern": np.array(pat)}
        for i, e in enumerate(embs):
            if e is not None:
                d[f"emb{i}"] = e
        for k, v in fus.state_dict().items():
            d["p." + k] = v
        for i, c in enumerate(cw):
            d[f"cw{i}"] = c
        for mode in ("eval", "train"):
            fus.train(mode == "train")
            fus.zero_grad()
            for e in embs:
                if e is not None:
                    e.grad = None
            with DropoutRecorder(torch.Generator().manual_seed(77 + ci)) as rec:
                outs = fus(embs)
            scalar = sum((o * c).sum() for o, c in zip(outs, cw))
            scalar.backward()
            for nm, o in zip(("joint", "joint_fz", "hidden", "weight_norm"), outs):
                d[f"{mode}.{nm}"] = o
            d[f"{mode}.scalar"] = scalar
            for i, e in enumerate(embs):
                if e is not None:
                    d[f"{mode}.g.emb{i}"] = e.grad
            for k, p in fus.named_parameters():
                if mode in pgrad_modes or k == "weight_raw":
                    d[f"{mode}.g.p.{k}"] = p.grad if p.grad is not None else torch.zeros_like(p)
            if mode == "train":
                # call order inside a layer: probs dropout, attention-output dropout, [FFN output dropout]
                per_layer = 3 if ffn else 2
                assert len(rec.masks) == per_layer * L, len(rec.masks)
                for l in range(L):
                    d[f"mask.{l}.probs"] = rec.masks[l * per_layer + 0]
                    d[f"mask.{l}.attn_out"] = rec.masks[l * per_layer + 1]
                    if ffn:
                        d[f"mask.{l}.ffn_out"] = rec.masks[l * per_layer + 2]
        save(f"g1_fusion_{name}.npz", d)


# --------------------------------------------------------------------------
def gen_g2():
    N, F_img, F_rel, F_att, D = 211, 48, 40, 24, 32
    g = torch.Generator().manual_seed(2024)
    img = F.normalize(torch.rand(N, F_img, generator=g))
    rel = torch.randn(N, F_rel, generator=g) * 2 + 0.5
    att = (torch.rand(N, F_att, generator=g) < 0.3).float()
    ent_wo_img = torch.tensor(sorted(set(torch.randint(0, N, (17,), generator=g).tolist())))
    ent_w = torch.randn(N, D, generator=g) / np.sqrt(N)
    args = types.SimpleNamespace(noise_ratio=0.2, mask_ratio=0.7)
    fake = types.SimpleNamespace(args=args, img_features=img, rel_features=rel, att_features=att,
                                 ent_wo_img=ent_wo_img,
                                 multimodal_encoder=types.SimpleNamespace(
                                     entity_emb=types.SimpleNamespace(weight=torch.nn.Parameter(ent_w.clone()))))
    fake.add_noise_to_embeddings = types.MethodType(SNAG.add_noise_to_embeddings, fake)
    SNAG.get_mean_std(fake)
    d = {"img": img, "rel": rel, "att": att, "ent_wo_img": ent_wo_img, "ent_w": ent_w,
         "noise_ratio": args.noise_ratio, "mask_ratio": args.mask_ratio,
         "img_mean": fake.img_mean, "img_std": fake.img_std, "rel_mean": fake.rel_mean, "rel_std": fake.rel_std,
         "att_mean": fake.att_mean, "att_std": fake.att_std}
    seed = 4242
    torch.manual_seed(seed)
    SNAG.update_noise(fake)
    d.update({"seed": seed, "rel_noisy": fake.rel_noisy_features, "att_noisy": fake.att_noisy_features,
              "img_noisy": fake.img_noisy_features, "ent_mean": fake.ent_mean, "ent_std": fake.ent_std,
              "entity_noise": fake.entity_noise, "entity_noise_mask": fake.entity_noise_mask})
    # replay of the reference's global-RNG call order (SNAG.py:68-70 x3, then :96, :98) -> explicit (mask, z)
    torch.manual_seed(seed)
    for nm, X in (("rel", rel), ("att", att), ("img", img)):
        mask = torch.rand(N) < args.noise_ratio
        z = torch.randn_like(X[mask])
        d[f"{nm}_mask"] = mask
        d[f"{nm}_z"] = z
    d["ent_z"] = torch.randn_like(ent_w)
    d["ent_u"] = torch.rand(N)
    assert torch.equal(d["ent_u"] < args.noise_ratio * 0.5, fake.entity_noise_mask)
    save("g2_noise.npz", d)


# --------------------------------------------------------------------------
def gen_g3():
    N, R, D, B = 157, 11, 128, 96
    g = torch.Generator().manual_seed(33)
    ent = torch.nn.Embedding(N, D)
    relm = torch.nn.Embedding(R, D)
    with torch.no_grad():
        ent.weight.copy_(torch.randn(N, D, generator=g) * 0.7)
        relm.weight.copy_(torch.randn(R, D, generator=g) * 0.2)
        ent.weight[5].zero_()     # exercises normalize eps
    zeros = lambda e: torch.zeros(e.shape[0], D)
    fake = types.SimpleNamespace(args=types.SimpleNamespace(margin=1.5), ent_embed=ent, rel_embed=relm,
                                 name_features=None, char_features=None)
    fake.r_rep = types.MethodType(MSNEA.r_rep, fake)
    fake.i_w = lambda e: zeros(e)                                   # term 2 becomes the constant n*margin
    fake._emb_generate = lambda e, a, b: (zeros(e), zeros(e), zeros(e), None, None)
    fake.fusion = lambda lst: lst[0]
    p_h = torch.randint(0, N, (B,), generator=g)
    p_r = torch.randint(0, R, (B,), generator=g)
    p_t = torch.randint(0, N, (B,), generator=g)
    p_h[0] = 5
    # in-tree sampler semantics: head-or-tail corruption, one negative per positive
    corrupt_head = torch.rand(B, generator=g) < 0.5
    n_h = torch.where(corrupt_head, torch.randint(0, N, (B,), generator=g), p_h)
    n_t = torch.where(corrupt_head, p_t, torch.randint(0, N, (B,), generator=g))
    n_r = p_r.clone()
    e1 = torch.arange(4)
    loss, *_ = MSNEA.kge(fake, p_h, p_r, p_t, n_h, n_r, n_t, e1, e1)
    loss.backward()
    term1 = loss.detach() - B * fake.args.margin                    # i_w == 0 -> pos1 = neg1 = 1 -> relu(margin)
    with torch.no_grad():
        pos = torch.sum(torch.square(fake.r_rep(p_h) + F.normalize(relm(p_r), 2, -1) - fake.r_rep(p_t)), dim=1)
        neg = torch.sum(torch.square(fake.r_rep(n_h) + F.normalize(relm(n_r), 2, -1) - fake.r_rep(n_t)), dim=1)
    save("g3_score.npz", {"ent": ent.weight, "rel": relm.weight, "p_h": p_h, "p_r": p_r, "p_t": p_t,
                          "n_h": n_h, "n_r": n_r, "n_t": n_t, "margin": fake.args.margin,
                          "loss_term1": term1, "loss_kge_total": loss, "pos": pos, "neg": neg,
                          "g.ent": ent.weight.grad, "g.rel": relm.weight.grad})


# --------------------------------------------------------------------------
class _ListHandler(logging.Handler):
    def __init__(self):
        super().__init__()
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def run_ref_test(final_emb, left, right, csls, csls_k):
    lg = logging.getLogger("golden")
    lg.setLevel(logging.INFO)
    lg.propagate = False
    h = _ListHandler()
    lg.handlers = [h]
    tmp = tempfile.mkdtemp()
    args = types.SimpleNamespace(model_name="SNAG", distance=2, csls=csls, csls_k=csls_k, data_path=tmp,
                                 data_choice="GOLD")
    model = types.SimpleNamespace(joint_emb_generat=lambda: (final_emb, None), state_dict=lambda: {})
    fake = types.SimpleNamespace(args=args, model=model, logger=lg, epoch=0, loss_item=0.0,
                                 early_stop_count=3, early_stop_init=3, loss_log=Loss_log())
    # Runner._test prints r2l only as rounded aggregates.  Its per-column ranks come from its OWN torch.sort calls (main.py:400-401
    # rows, :422-423 columns): record every sort it makes, nothing is recomputed here.
    sorted_indices = []
    real_sort = torch.sort

    def recording_sort(*a, **k):
        out = real_sort(*a, **k)
        sorted_indices.append(out[1].clone())
        return out

    torch.sort = recording_sort
    try:
        ref_main.Runner._test(fake, left, right, last_epoch=True, save_name="g")
    finally:
        torch.sort = real_sort
    n_l, n_r = len(left), len(right)
    assert len(sorted_indices) == n_l + n_r, (len(sorted_indices), n_l, n_r)
    ranks = []
    with open(os.path.join(tmp, "SNAG", "g_pred", "GOLD_pred.txt")) as f:
        rd = csv.reader(f)
        next(rd)
        for row in rd:
            ranks.append(int(row[1]))
    out = {"rank_l2r": np.array(ranks)}
    sort_l2r = np.array([int((ind == i).nonzero(as_tuple=False).squeeze().item()) for i, ind in enumerate(sorted_indices[:n_l])])
    assert np.array_equal(sort_l2r, out["rank_l2r"])                 # the recorded sorts ARE the ones behind the csv ranks
    out["rank_r2l"] = np.array([int((ind == i).nonzero(as_tuple=False).squeeze().item()) for i, ind in enumerate(sorted_indices[n_l:])])
    for name, rk in (("l2r", out["rank_l2r"]), ("r2l", out["rank_r2l"])):
        mr = mrr = 0.0                                               # the reference's own accumulation order (main.py:403-404,425-426,430-433)
        for v in rk.tolist():
            mr += (v + 1)
            mrr += 1.0 / (v + 1)
        out[f"mr_{name}"] = mr / len(rk)
        out[f"mrr_{name}"] = mrr / len(rk)
    for line in h.lines:
        m = re.search(r"\| (l2r|r2l): acc of top \[1, 10, 50\] = \[([^\]]*)\], mr = ([\d.]+), mrr = ([\d.]+)", line)
        if m:
            out[f"hits_{m.group(1)}"] = np.array([float(x) for x in m.group(2).split()], dtype=np.float64)
            out[f"mr_{m.group(1)}_3dp"] = float(m.group(3))
            out[f"mrr_{m.group(1)}_3dp"] = float(m.group(4))
    assert "hits_l2r" in out and "hits_r2l" in out, h.lines
    for name in ("l2r", "r2l"):                                      # the unrounded values reproduce the log line's 3 decimals
        assert f"{out[f'mr_{name}']:.3f}" == f"{out[f'mr_{name}_3dp']:.3f}" and f"{out[f'mrr_{name}']:.3f}" == f"{out[f'mrr_{name}_3dp']:.3f}"
    return out


def gen_g4_g5():
    for ci, (name, n, dim, csls_k, sigma) in enumerate((("d64", 300, 64, 3, 2.4), ("d1200", 120, 1200, 3, 11.0),
                                                        ("d64_nocsls", 200, 64, 0, 2.2))):
        g = torch.Generator().manual_seed(400 + ci)
        total = 2 * n + 40
        left_rows = torch.randn(n, dim, generator=g)
        right_rows = left_rows + sigma * torch.randn(n, dim, generator=g)      # planted signal: hits@1 in (0,1)
        extra = torch.randn(40, dim, generator=g)
        emb = torch.cat([left_rows, right_rows, extra], 0) * 3.0              # un-normalised on purpose
        perm = torch.randperm(total, generator=g)
        emb = emb[perm]
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(total)
        left = inv[:n].clone()
        right = inv[n:2 * n].clone()
        final = F.normalize(emb)
        dist = pairwise_distances(final[left], final[right])
        d = {"emb": emb, "left": left, "right": right, "csls_k": csls_k, "dist": dist}
        if csls_k:
            d["csls"] = csls_sim(1 - dist, csls_k)
        d.update(run_ref_test(emb, left, right, bool(csls_k), csls_k))
        save(f"g4_eval_{name}.npz", d)

    # G5: Iter_new_links (mutual nearest neighbours) with >1 chunk of 1000 rows
    g = torch.Generator().manual_seed(55)
    nl, nr, dim = 1300, 1100, 48
    base = torch.randn(max(nl, nr), dim, generator=g)
    final = F.normalize(torch.cat([base[:nl] + 0.6 * torch.randn(nl, dim, generator=g),
                                   base[:nr] + 0.6 * torch.randn(nr, dim, generator=g)], 0))
    left = list(range(nl))
    right = list(range(nl, nl + nr))
    fake = types.SimpleNamespace(args=types.SimpleNamespace(semi_learn_step=5))
    links = SNAG.Iter_new_links(fake, 4, left, final, right, new_links=[])   # (4+1) % 25 == 5 -> first branch
    save("g5_il.npz", {"final": final, "left": np.array(left), "right": np.array(right),
                       "links": np.array(links, dtype=np.int64)})


# --------------------------------------------------------------------------
def gen_g6():
    g = torch.Generator().manual_seed(66)
    N, Fin, D = 70, 200, 128
    x = torch.randn(N, Fin, generator=g)
    lin = torch.nn.Linear(Fin, D)
    randomize_(lin, g)
    cw = torch.randn(N, D, generator=g)
    y = lin(x)
    (y * cw).sum().backward()
    d = {"x": x, "lin.weight": lin.weight, "lin.bias": lin.bias, "cw": cw, "lin.y": y,
         "lin.g.weight": lin.weight.grad, "lin.g.bias": lin.bias.grad}
    ph = ProjectionHead(Fin, 96, D, dropout=0.25)
    randomize_(ph, g)
    xr = x.clone().requires_grad_(True)
    for mode in ("eval", "train"):
        ph.train(mode == "train")
        ph.zero_grad()
        xr.grad = None
        with DropoutRecorder(torch.Generator().manual_seed(9)) as rec:
            y2 = ph(xr)
        (y2 * cw).sum().backward()
        d[f"ph.{mode}.y"] = y2
        d[f"ph.{mode}.g.l1"] = ph.l1.weight.grad
        d[f"ph.{mode}.g.l2"] = ph.l2.weight.grad
        d[f"ph.{mode}.g.x"] = xr.grad
        if mode == "train":
            d["ph.mask"] = rec.masks[0]
    d.update({"ph.l1": ph.l1.weight, "ph.l2": ph.l2.weight, "ph.p": 0.25})
    save("g6_proj.npz", d)


# --------------------------------------------------------------------------
def gen_g7():
    """MultiModalEncoder.forward with the structural encoder stubbed to identity (GAT is out of scope)."""
    g = torch.Generator().manual_seed(707)
    N, D, F_img, F_att = 61, 128, 96, 80
    args = types.SimpleNamespace(
        attr_dim=D, img_dim=D, name_dim=D, char_dim=D, dropout=0.0, hidden_units=f"{D},{D},{D}", heads="2,2",
        structure_encoder="gcn", attn_dropout=0.0, instance_normalization=False, inner_view_num=4, with_weight=1,
        w_gcn=True, w_img=True, w_rel=True, w_attr=True, w_name=False, w_char=False, mask_ratio=0.7,
        hidden_size=D, num_attention_heads=2, num_hidden_layers=1, use_intermediate=0, intermediate_size=400)
    enc = MultiModalEncoder(args, ent_num=N, img_feature_dim=F_img, char_feature_dim=100, attr_input_dim=F_att)
    randomize_(enc, g)
    with torch.no_grad():
        enc.entity_emb.weight.copy_(torch.randn(N, D, generator=g) / np.sqrt(N))
    class _Identity(torch.nn.Module):
        def forward(self, x, adj):
            return x
    enc.cross_graph_model = _Identity()
    enc.eval()
    img = F.normalize(torch.rand(N, F_img, generator=g))
    rel = torch.randn(N, 1000, generator=g) * (torch.rand(N, 1000, generator=g) < 0.05)
    att = (torch.rand(N, F_att, generator=g) < 0.2).float()
    noise = torch.randn(N, D, generator=g) * 0.1
    nmask = torch.rand(N, generator=g) < 0.3
    idx = torch.arange(N)
    d = {"img": img, "rel": rel, "att": att, "entity_noise": noise, "entity_noise_mask": nmask, "mask_ratio": 0.7}
    for k, v in enc.state_dict().items():
        if k.startswith(("entity_emb", "rel_fc", "att_fc", "img_fc", "fusion")):
            d["p." + k] = v
    names = ("gph", "img", "rel", "att", "name", "char", "joint", "joint_fz", "hidden", "weight_norm")
    for tag, kw in (("clean", {}), ("noisy", {"entity_noise": noise, "entity_noise_mask": nmask})):
        enc.zero_grad()
        outs = enc(idx, None, img, rel, att, None, None, **kw)
        for nm, o in zip(names, outs):
            if o is not None:
                d[f"{tag}.{nm}"] = o
        # gradient of sum(hidden[:, graph slot]) w.r.t. the entity table (checks the (1 - eps/2) scaling)
        outs[8][:, 3, :].sum().backward()
        d[f"{tag}.g.entity_emb"] = enc.entity_emb.weight.grad.clone()
    save("g7_encoder.npz", d)


def gen_g8():
    """icl_loss (model/SNAG_loss.py:31-128): bidirectional InfoNCE over in-batch pairs, optional per-entity weight."""
    from model.SNAG_loss import icl_loss, CustomMultiLossLayer
    g = torch.Generator().manual_seed(88)
    N, d, B = 260, 96, 70
    emb = (torch.randn(N, d, generator=g) * 1.7).requires_grad_(True)
    perm = torch.randperm(N, generator=g)
    links = torch.stack([perm[:B], perm[B:2 * B]], dim=1).numpy().astype(np.int32)
    wn = torch.rand(N, generator=g) * 3
    out = {"emb": emb, "links": links, "weight_norm": wn}
    for tag, tau, ab, w in (("plain", 0.1, 0.5, None), ("weighted", 0.1, 0.5, wn), ("asym", 0.05, 0.3, None)):
        crit = icl_loss(tau=tau, ab_weight=ab, n_view=2)
        emb.grad = None
        loss = crit(emb, links, weight_norm=w)
        loss.backward()
        out[f"{tag}.tau"], out[f"{tag}.ab"] = tau, ab
        out[f"{tag}.loss"], out[f"{tag}.g.emb"] = loss, emb.grad.clone()
    # CustomMultiLossLayer (model/SNAG_loss.py:12-29)
    ml = CustomMultiLossLayer(loss_num=6)
    with torch.no_grad():
        ml.log_vars.copy_(torch.randn(6, generator=g) * 0.3)
    ls = [torch.tensor(float(x), requires_grad=True) for x in (1.5, 0.2, 3.0, 0.7)]
    tot = ml(ls)
    tot.backward()
    out["ml.log_vars"], out["ml.losses"], out["ml.total"] = ml.log_vars, torch.stack([l.detach() for l in ls]), tot
    out["ml.g.log_vars"], out["ml.g.losses"] = ml.log_vars.grad, torch.stack([l.grad for l in ls])
    save("g8_icl_loss.npz", out)


def gen_g9():
    """Sparse GAT (diag) as SNAG builds it: GAT(n_units, n_heads, dropout=0, attn_dropout=0, instance_normalization=False, diag=True)."""
    from model.Tool_model import GAT
    from src.utils import get_adjr
    g = torch.Generator().manual_seed(99)
    N, f = 180, 64
    triples = [(int(a), 0, int(b)) for a, b in zip(torch.randint(0, N, (700,), generator=g), torch.randint(0, N, (700,), generator=g))]
    adj = get_adjr(N, triples, norm=True)
    gat = GAT(n_units=[f, f, f], n_heads=[2, 2], dropout=0.0, attn_dropout=0.0, instance_normalization=False, diag=True)
    randomize_(gat, g)
    with torch.no_grad():
        for l in gat.layer_stack:
            l.w.copy_(1.0 + 0.3 * torch.randn(l.w.shape, generator=g))
            l.a_src_dst.copy_(torch.randn(l.a_src_dst.shape, generator=g) * 0.4)
    gat.eval()
    x = (torch.randn(N, f, generator=g) * 0.8).requires_grad_(True)
    cw = torch.randn(N, f, generator=g)
    y = gat(x, adj)
    (y * cw).sum().backward()
    d = {"x": x, "edges": adj._indices(), "cw": cw, "y": y, "g.x": x.grad, "n_heads": np.array([2, 2])}
    for k, v in gat.state_dict().items():
        d["p." + k] = v
    for k, p in gat.named_parameters():
        d["g.p." + k] = p.grad
    save("g9_gat.npz", d)


def gen_g12():
    """GCN structural encoder option: GCN(nfeat, nhid, nout, dropout=0) over the normalised adjacency (values used by torch.spmm)."""
    from model.Tool_model import GCN
    from src.utils import get_adjr
    g = torch.Generator().manual_seed(1212)
    N, f = 150, 48
    triples = [(int(a), 0, int(b)) for a, b in zip(torch.randint(0, N, (600,), generator=g), torch.randint(0, N, (600,), generator=g))]
    adj = get_adjr(N, triples, norm=True)
    gcn = GCN(f, 40, 36, dropout=0.0)
    randomize_(gcn, g, scale=0.5)
    gcn.eval()
    x = (torch.randn(N, f, generator=g) * 0.8).requires_grad_(True)
    cw = torch.randn(N, 36, generator=g)
    y = gcn(x, adj)
    (y * cw).sum().backward()
    d = {"x": x, "adj_indices": adj._indices(), "adj_values": adj._values(), "cw": cw, "y": y, "g.x": x.grad}
    for k, v in gcn.state_dict().items():
        d["p." + k] = v
    for k, p in gcn.named_parameters():
        d["g.p." + k] = p.grad
    save("g12_gcn.npz", d)


def snag_args(D=64, add_noise=1):
    """The flags SNAG / MultiModalEncoder / MformerFusion read (config.py:16-141), run_snag.sh values scaled down to D."""
    return types.SimpleNamespace(
        attr_dim=D, img_dim=D, name_dim=D, char_dim=D, dropout=0.0, attn_dropout=0.0, hidden_units=f"{D},{D},{D}", heads="2,2",
        structure_encoder="gat", instance_normalization=False, inner_view_num=4, with_weight=1,
        w_gcn=True, w_img=True, w_rel=True, w_attr=True, w_name=False, w_char=False, use_project_head=False,
        hidden_size=D, num_attention_heads=1, num_hidden_layers=1, use_intermediate=1, intermediate_size=96,
        add_noise=add_noise, noise_ratio=0.2, mask_ratio=0.7, tau=0.1, tau2=4.0, ab_weight=0.5, neg_cross_kg=0, zoom=0.1,
        reduction="mean", batch_size=60, awloss=0, semi_learn_step=5, modal_num=3)


def gen_g10():
    """The whole in-tree model: SNAG.forward (model/SNAG.py:101-122) on a synthetic KG, eval-mode fusion dropout avoided by
    recording the Bert dropout masks; loss, the three loss groups, joint embeddings and gradients."""
    from src.utils import get_adjr
    g = torch.Generator().manual_seed(1010)
    N, D, F_img, F_att, B = 240, 64, 80, 56, 60
    args = snag_args(D)
    triples = [(int(a), 0, int(b)) for a, b in zip(torch.randint(0, N, (900,), generator=g), torch.randint(0, N, (900,), generator=g))]
    perm = torch.randperm(N, generator=g)
    train_ill = torch.stack([perm[:B], perm[B:2 * B]], 1).numpy().astype(np.int32)
    ent_wo_img = sorted(set(torch.randint(0, N, (20,), generator=g).tolist()))
    kgs = {"ent_num": N, "images_list": torch.rand(N, F_img, generator=g).numpy(),
           "rel_features": (torch.rand(N, 1000, generator=g) < 0.02).float().numpy(),
           "att_features": (torch.rand(N, F_att, generator=g) < 0.2).float().numpy(),
           "name_features": None, "char_features": None, "input_idx": torch.arange(N),
           "ent_wo_img": ent_wo_img, "ent_w_img": [i for i in range(N) if i not in ent_wo_img],
           "adj": get_adjr(N, triples, norm=True), "train_ill": train_ill}
    torch.manual_seed(7)
    model = SNAG(kgs, args)
    randomize_(model.multimodal_encoder.fusion, g)
    with torch.no_grad():
        model.multi_loss_layer.log_vars.copy_(torch.randn(6, generator=g) * 0.2)
        for l in model.multimodal_encoder.cross_graph_model.layer_stack:
            l.w.copy_(1.0 + 0.2 * torch.randn(l.w.shape, generator=g))
    d = {"N": N, "D": D, "train_ill": train_ill, "edges": kgs["adj"]._indices(), "images_list": kgs["images_list"],
         "rel_features": kgs["rel_features"], "att_features": kgs["att_features"], "ent_wo_img": np.array(ent_wo_img)}
    for k, v in model.state_dict().items():
        d["p." + k] = v
    # eval-mode (clean tables, no dropout): joint embedding used by Runner._test
    model.eval()
    with torch.no_grad():
        final, wn = model.joint_emb_generat()
    d["eval.joint_emb_fz"], d["eval.weight_norm"] = final, wn
    # train mode: noise from a seeded global RNG + recorded Bert dropout masks
    model.train()
    torch.manual_seed(4321)
    model.update_noise()
    d["noise_seed"] = 4321
    d["img_noisy"], d["rel_noisy"], d["att_noisy"] = model.img_noisy_features, model.rel_noisy_features, model.att_noisy_features
    d["entity_noise"], d["entity_noise_mask"] = model.entity_noise, model.entity_noise_mask
    with DropoutRecorder(torch.Generator().manual_seed(5)) as rec:
        loss, out = model(train_ill)
    loss.backward()
    assert len(rec.masks) == 3, len(rec.masks)
    d["mask.probs"], d["mask.attn_out"], d["mask.ffn_out"] = rec.masks
    d["train.loss"] = loss
    for k, v in out["loss_dic"].items():
        d["train.loss." + k] = v
    d["train.joint_emb"] = out["emb"]
    for k, p in model.named_parameters():
        if p.grad is not None:
            d["g.p." + k] = p.grad
    save("g10_snag_model.npz", d)


def gen_g11():
    import pickle
    import src.data as ref_data
    from src.utils import get_adjr
    rng = np.random.RandomState(2024)
    n1, n2, n_rel, F = 40, 45, 12, 16
    N = n1 + n2
    files = {}
    files["ent_ids_1"] = "".join(f"{i}\thttp://kg1/e{i}\n" for i in range(n1))
    files["ent_ids_2"] = "".join(f"{i}\thttp://kg2/e{i}\n" for i in range(n1, N))

    def triples(lo, hi, n):
        rows = []
        for _ in range(n):
            h, t = rng.randint(lo, hi, 2)
            rows.append((h, rng.randint(0, n_rel) if rng.rand() < 0.8 else 3, t))     # relation 3 is frequent, some h == t occur
        rows += [(rows[0][2], rows[0][1], rows[0][0]), rows[1], rows[1]]               # a reverse pair and duplicates
        return "".join(f"{h}\t{r}\t{t}\n" for h, r, t in rows)
    files["triples_1"] = triples(0, n1, 150)
    files["triples_2"] = triples(n1, N, 160)
    left = rng.permutation(n1)[:30]
    right = n1 + rng.permutation(n2)[:30]
    files["ill_ent_ids"] = "".join(f"{a}\t{b}\n" for a, b in zip(left, right))
    for k, (lo, hi, tag) in enumerate(((0, n1, "kg1"), (n1, N, "kg2")), 1):
        lines = []
        for i in range(lo, hi):
            if rng.rand() < 0.85:
                attrs = [f"attr{a}" for a in rng.randint(0, 30, rng.randint(1, 9))]
                lines.append("\t".join([f"http://{tag}/e{i}"] + attrs))
        lines.append("http://elsewhere/unknown\tattr1\tattr2")                       # a name that is not in ent2id
        files[f"training_attrs_{k}"] = "\n".join(lines) + "\n"
    img_ids = rng.permutation(N)[:60]
    img_vecs = rng.randn(60, F).astype(np.float32)

    with tempfile.TemporaryDirectory() as root:
        file_dir = os.path.join(root, "DBP15K", "zh_en")
        os.makedirs(file_dir)
        os.makedirs(os.path.join(root, "pkls"))
        for name, text in files.items():
            with open(os.path.join(file_dir, name), "w", encoding="utf-8") as f:
                f.write(text)
        with open(os.path.join(root, "pkls", "zh_en_GA_id_img_feature_dict.pkl"), "wb") as f:
            pickle.dump({int(i): v for i, v in zip(img_ids, img_vecs)}, f)
        ent2id, ills, tri, r_hs, r_ts, ids = ref_data.read_raw_data(file_dir, [1, 2])
        d = {"file_names": np.array(sorted(files)), "img_ids": img_ids, "img_vecs": img_vecs,
             "raw_ills": np.array(ills), "raw_triples": np.array(tri), "n_rel": len(r_hs),
             "ent_names": np.array(sorted(ent2id)), "ent_ids": np.array([ent2id[k] for k in sorted(ent2id)]),
             "left_ids": np.array(ref_data.get_ids(os.path.join(file_dir, "ent_ids_1")))}
        for name in files:
            d["file." + name] = np.frombuffer(files[name].encode("utf-8"), dtype=np.uint8)
        adj0 = get_adjr(N, tri, norm=False)
        d["adj0_indices"], d["adj0_values"] = adj0._indices(), adj0._values()
        args = types.SimpleNamespace(data_path=root, data_choice="DBP15K", data_split="zh_en", ratio="1.0", data_rate=0.3, word_embedding="glove",
                                     w_name=0, w_char=0, unsup=0)
        np.random.seed(123)
        kgs, non_train, train_ds, test_ds, eval_ill, test_ill_ = ref_data.load_eva_data(logging.getLogger("g11"), args)
        d.update({"seed": 123, "data_rate": 0.3, "ent_num": kgs["ent_num"], "rel_num": kgs["rel_num"], "images_list": kgs["images_list"],
                  "rel_features": kgs["rel_features"], "att_features": kgs["att_features"], "ent_wo_img": np.array(kgs["ent_wo_img"]),
                  "ent_w_img": np.array(kgs["ent_w_img"]), "adj_indices": kgs["adj"]._indices(), "adj_values": kgs["adj"]._values(),
                  "train_ill": kgs["train_ill"], "test_ill": np.array(test_ill_), "left_non_train": np.array(non_train["left"]),
                  "right_non_train": np.array(non_train["right"]), "input_idx": kgs["input_idx"]})
    save("g11_data.npz", d)


def gen_g14():
    """set_optim (src/utils.py:25-86) on the reference SNAG model of golden G10: which parameter NAMES land in which group, with
    which lr / weight decay, and the cosine schedule's first learning rates."""
    from src.utils import set_optim
    z = dict(np.load(os.path.join(OUT, "g10_snag_model.npz")))
    N = int(z["N"])
    adj = torch.sparse_coo_tensor(torch.from_numpy(z["edges"]), torch.ones(z["edges"].shape[1]), (N, N))
    wo = z["ent_wo_img"].tolist()
    kgs = {"ent_num": N, "images_list": z["images_list"], "rel_features": z["rel_features"], "att_features": z["att_features"],
           "name_features": None, "char_features": None, "input_idx": torch.arange(N), "ent_wo_img": wo,
           "ent_w_img": [i for i in range(N) if i not in set(wo)], "adj": adj, "train_ill": z["train_ill"]}
    model = SNAG(kgs, snag_args(int(z["D"])))
    opt = types.SimpleNamespace(model_name="SNAG", lr=5e-4, weight_decay=1e-4, adam_epsilon=1e-8, optim="adamw", scheduler="cos",
                                warmup_steps=40, total_steps=400, accumulation_steps=2)
    optimizer, scheduler = set_optim(opt, [model], [], [], None)
    name_of = {id(p): n for n, p in model.named_parameters()}
    d = {}
    for gi, gr in enumerate(optimizer.param_groups):
        d[f"group{gi}.names"] = np.array([name_of[id(p)] for p in gr["params"]])
        d[f"group{gi}.initial_lr"] = gr["initial_lr"]
        d[f"group{gi}.weight_decay"] = gr["weight_decay"]
    lrs = []
    for _ in range(30):
        lrs.append([g_["lr"] for g_ in optimizer.param_groups])
        optimizer.step()
        scheduler.step()
    d["lrs"] = np.array(lrs)
    save("g14_set_optim.npz", d)


def gen_g13():
    """Surface-form features and the unsupervised split of the loader, on golden G11's synthetic dataset plus a synthetic 300-d GloVe
    file and translated-name JSON: load_word_char_features / load_eva_data(w_name, w_char) / load_eva_data(unsup) (src/data.py)."""
    import json
    import pickle
    import src.data as ref_data
    z = dict(np.load(os.path.join(OUT, "g11_data.npz")))
    rng = np.random.RandomState(77)
    N = int(z["ent_num"])
    vocab = [f"w{i}" for i in range(40)]
    glove = "".join(w + " " + " ".join(f"{v:.5f}" for v in rng.randn(300)) + "\n" for w in vocab)
    glove += "badline 1.0 2.0\n" + "w_nan " + " ".join(["x"] * 300) + "\n"                 # wrong width, unparsable: both skipped
    names = []
    for i in range(N):
        kind = i % 9
        if kind == 0:
            words = ["zzunknown"]                                 # no GloVe hit: random name vector; still has bigrams
        elif kind == 1:
            words = ["q"]                                         # no hit and no bigram: both random
        else:
            words = [vocab[j] for j in rng.randint(0, 40, rng.randint(1, 4))] + (["Mixed" + vocab[i % 40].upper()] if kind == 2 else [])
        names.append([i, words])
    names_json = json.dumps(names)
    with tempfile.TemporaryDirectory() as root:
        file_dir = os.path.join(root, "DBP15K", "zh_en")
        os.makedirs(file_dir)
        os.makedirs(os.path.join(root, "pkls"))
        os.makedirs(os.path.join(root, "embedding"))
        os.makedirs(os.path.join(root, "DBP15K", "translated_ent_name"))
        for name in z["file_names"]:
            with open(os.path.join(file_dir, str(name)), "wb") as f:
                f.write(z["file." + str(name)].tobytes())
        with open(os.path.join(root, "pkls", "zh_en_GA_id_img_feature_dict.pkl"), "wb") as f:
            pickle.dump({int(i): v for i, v in zip(z["img_ids"], z["img_vecs"])}, f)
        with open(os.path.join(root, "embedding", "glove.6B.300d.txt"), "w", encoding="utf-8") as f:
            f.write(glove)
        with open(os.path.join(root, "DBP15K", "translated_ent_name", "dbp_zh_en.json"), "w") as f:
            f.write(names_json)
        args = types.SimpleNamespace(data_path=root, data_choice="DBP15K", data_split="zh_en", ratio="1.0", data_rate=0.3, word_embedding="glove",
                                     w_name=1, w_char=1, unsup=0, unsup_mode="img", unsup_k=10)
        np.random.seed(321)
        kgs, non_train, _, _, _, test_ill_ = ref_data.load_eva_data(logging.getLogger("g13"), args)
        d = {"glove": np.frombuffer(glove.encode("utf-8"), dtype=np.uint8), "names_json": np.frombuffer(names_json.encode("utf-8"), dtype=np.uint8),
             "seed": 321, "name_features": kgs["name_features"], "char_features": kgs["char_features"], "images_list": kgs["images_list"],
             "train_ill": kgs["train_ill"], "cached_name": pickle.load(open(os.path.join(root, "embedding", "dbp_zh_en_name.pkl"), "rb")),
             "cached_char": pickle.load(open(os.path.join(root, "embedding", "dbp_zh_en_char.pkl"), "rb"))}
        for mode in ("img", "name", "char"):
            args.unsup, args.unsup_mode = 1, mode
            np.random.seed(321)
            kgs_u, non_u, _, _, _, test_u = ref_data.load_eva_data(logging.getLogger("g13"), args)
            d[f"unsup_{mode}.train_ill"] = kgs_u["train_ill"]
            d[f"unsup_{mode}.test_ill"] = np.array(test_u)
            d[f"unsup_{mode}.left_non_train"] = np.array(non_u["left"])
    save("g13_data_surface_unsup.npz", d)


def gen_g16():
    """GAT training options: input dropout of the non-final layers (Tool_model.py:78-79) and edge dropout after the row sums (layers.py:83).
    Masks in call order: layer 0 input, layer 0 head 0 edges, head 1 edges, layer 1 head 0 edges, head 1 edges."""
    from model.Tool_model import GAT
    from src.utils import get_adjr
    g = torch.Generator().manual_seed(1616)
    N, f = 150, 64
    triples = [(int(a), 0, int(b)) for a, b in zip(torch.randint(0, N, (600,), generator=g), torch.randint(0, N, (600,), generator=g))]
    adj = get_adjr(N, triples, norm=True)
    gat = GAT(n_units=[f, f, f], n_heads=[2, 2], dropout=0.2, attn_dropout=0.3, instance_normalization=False, diag=True)
    with torch.no_grad():
        for l in gat.layer_stack:
            l.w.copy_(1.0 + 0.3 * torch.randn(l.w.shape, generator=g))
            l.a_src_dst.copy_(torch.randn(l.a_src_dst.shape, generator=g) * 0.4)
    gat.train()
    x = (torch.randn(N, f, generator=g) * 0.8).requires_grad_(True)
    cw = torch.randn(N, f, generator=g)
    with DropoutRecorder(torch.Generator().manual_seed(16)) as rec:
        y = gat(x, adj)
    (y * cw).sum().backward()
    assert len(rec.masks) == 5 and tuple(rec.masks[0].shape) == (N, f)
    d = {"x": x, "edges": adj._indices(), "cw": cw, "y": y, "g.x": x.grad, "n_heads": np.array([2, 2]), "p_in": 0.2, "p_edge": 0.3,
         "in_keep0": rec.masks[0], "edge_keep0": torch.stack(rec.masks[1:3]), "edge_keep1": torch.stack(rec.masks[3:5])}
    for k, v in gat.state_dict().items():
        d["p." + k] = v
    for k, p in gat.named_parameters():
        d["g.p." + k] = p.grad
    save("g16_gat_dropout.npz", d)


def gen_g15():
    """The non-transformer attention read-out: SelfAttention(hidden, batch_first=True) on [N, M, D] tokens, M in {3, 4, 6}, D in {128, 300}."""
    import warnings
    from model.EVA_tools import SelfAttention
    g = torch.Generator().manual_seed(1515)
    d = {}
    for tag, (N, M, D) in {"m3_d128": (67, 3, 128), "m4_d300": (33, 4, 300), "m6_d64": (21, 6, 64)}.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # nn.init.xavier_uniform / implicit softmax dim are deprecated spellings
            att = SelfAttention(D, batch_first=True)
            att.att_weights.data = torch.randn(1, D, generator=g) * 0.3
            x = (torch.randn(N, M, D, generator=g)).requires_grad_(True)
            cw = torch.randn(N, D, generator=g)
            rep, w = att(x)
        (rep * cw).sum().backward()
        d.update({f"{tag}.x": x, f"{tag}.a": att.att_weights.data[0], f"{tag}.cw": cw, f"{tag}.rep": rep, f"{tag}.w": w,
                  f"{tag}.g.x": x.grad, f"{tag}.g.a": att.att_weights.grad[0]})
    save("g15_atten_weight.npz", d)


if __name__ == "__main__":
    if len(sys.argv) > 1:          # regenerate selected vectors only: gen_golden.py g11 g8 ...
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    for fn in (gen_g1, gen_g2, gen_g3, gen_g4_g5, gen_g6, gen_g7, gen_g8, gen_g9, gen_g10, gen_g11, gen_g12, gen_g13, gen_g14, gen_g15, gen_g16):
        fn()                        # "regenerate all" means ALL: G11-G14 included
