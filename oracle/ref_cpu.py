"""CPU oracle for the SNAG fusion + negative-sampled scoring hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``snag_amd``) never imports anything from
``oracle/`` and has no CPU fallback.

It is a plain-PyTorch fp32 restatement (no reference imports, no reference
source) of the reference's algorithms, written from SURVEY.md Appendix C.  Every
function cites the reference file:line it follows (paths relative to
``/root/reference/SNAG_MMEA``).  Because everything is ordinary differentiable
torch code, ``torch.autograd`` of these functions is the backward oracle too.

Parity status
-------------
* PINNED against the reference itself (imported in the build container, vectors
  committed under ``tests/golden/`` by ``tests/golden/gen_golden.py``):
  fusion forward/backward (a1-a4), projections (a5), noise injection (a6),
  in-tree TransE-L2^2 margin scorer ``MSNEA.kge`` term 1 (a7), similarity
  eval / CSLS / Hits@k / mutual-NN (a8).
* PARITY UNPINNED (the MKGC source is absent from the reference tree, SURVEY.md
  par. 0.1): the K>1 "one positive vs K negative tails" pairing of the margin
  loss, and the ``Mformer_hd_mean`` / ``Mformer_hd_graph`` read-out names.  They
  follow BASELINE.json's north_star text; their fixtures are self-generated.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------
# small building blocks
# --------------------------------------------------------------------------


def nrm(v: Tensor) -> Tensor:
    """``F.normalize(v)`` : v / max(||v||_2, 1e-12) over dim 1 (last dim for 2-D).

    reference: model/SNAG_tools.py:44,48 ; model/MSNEA.py:151-152
    """
    n = v.pow(2).sum(dim=-1, keepdim=True).sqrt().clamp_min(1e-12)
    return v / n


def layer_norm(v: Tensor, g: Tensor, b: Tensor, eps: float = 1e-12) -> Tensor:
    """nn.LayerNorm over the last dim, biased variance, eps inside the sqrt.

    reference: model/SNAG_tools.py:215,221,259,265
    """
    mu = v.mean(dim=-1, keepdim=True)
    var = (v - mu).pow(2).mean(dim=-1, keepdim=True)
    return (v - mu) / torch.sqrt(var + eps) * g + b


def gelu_erf(x: Tensor) -> Tensor:
    """ACT2FN["gelu"] == exact erf GELU.  reference: model/SNAG_tools.py:248"""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """nn.Linear: x @ w.T + b with w[out, in].  reference: model/SNAG_tools.py:92-96,134-150"""
    y = x @ w.t()
    if b is not None:
        y = y + b
    return y


def projection_head(x: Tensor, w1: Tensor, w2: Tensor,
                    keep_mask: Optional[Tensor] = None, p: float = 0.0) -> Tensor:
    """ProjectionHead: Linear(no bias) -> ReLU -> dropout -> Linear(no bias).

    reference: model/layers.py:135-148 (wired in model/MCLEA_tools.py:96-100).
    ``keep_mask`` (0/1, same shape as the hidden activation) makes dropout
    explicit: h * keep / (1 - p).
    """
    h = torch.relu(x @ w1.t())
    if keep_mask is not None:
        h = h * keep_mask.to(h.dtype) / (1.0 - p)
    return h @ w2.t()


# --------------------------------------------------------------------------
# a1-a4, a9: Mformer fusion
# --------------------------------------------------------------------------


def _apply_keep(x: Tensor, keep: Optional[Tensor], p: float) -> Tensor:
    if keep is None:
        return x
    return x * keep.to(x.dtype) / (1.0 - p)


def fusion_forward(embs: Sequence[Optional[Tensor]],
                   params: Dict[str, Tensor],
                   num_heads: int,
                   num_layers: int = 1,
                   use_intermediate: bool = False,
                   drop_masks: Optional[List[Dict[str, Tensor]]] = None,
                   drop_p: float = 0.1,
                   ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """MformerFusion.forward.

    reference: model/SNAG_tools.py:32-51 (fusion), 178-209 (self-attention),
    218-222 (self-output), 244-266 (FFN), 279-298 (layer).

    ``params`` uses the reference's state_dict keys (``weight_raw``,
    ``fusion_layer.{l}.attention.self.query.weight`` ...).  ``drop_masks`` is
    ``None`` for eval mode, else one dict per layer with 0/1 keep masks
    ``probs`` [N,H,M,M], ``attn_out`` [N,M,D] and (FFN only) ``ffn_out`` [N,M,D]
    (inverted dropout, p = 0.1, reference lines 169,201,216,220,260,264).
    Returns (joint_emb, joint_emb_fz, hidden_states, weight_norm).
    """
    embs = [e for e in embs if e is not None]                       # :33
    M = len(embs)
    x = torch.stack(embs, dim=1)                                    # :36  [N,M,D]
    N, _, D = x.shape
    H = num_heads
    d = D // H
    probs = None
    for l in range(num_layers):
        pre = f"fusion_layer.{l}."
        dm = drop_masks[l] if drop_masks is not None else {}
        q = linear(x, params[pre + "attention.self.query.weight"], params[pre + "attention.self.query.bias"])
        k = linear(x, params[pre + "attention.self.key.weight"], params[pre + "attention.self.key.bias"])
        v = linear(x, params[pre + "attention.self.value.weight"], params[pre + "attention.self.value.bias"])
        q = q.view(N, M, H, d).permute(0, 2, 1, 3)                   # :171-176
        k = k.view(N, M, H, d).permute(0, 2, 1, 3)
        v = v.view(N, M, H, d).permute(0, 2, 1, 3)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(d)                 # :192-193
        probs = torch.softmax(s, dim=-1)                             # :196
        probs = _apply_keep(probs, dm.get("probs"), drop_p)          # :201
        ctx = (probs @ v).permute(0, 2, 1, 3).reshape(N, M, D)       # :202-206
        o = linear(ctx, params[pre + "attention.output.dense.weight"], params[pre + "attention.output.dense.bias"])
        o = _apply_keep(o, dm.get("attn_out"), drop_p)               # :220
        y = layer_norm(o + x, params[pre + "attention.output.LayerNorm.weight"],
                       params[pre + "attention.output.LayerNorm.bias"])   # :221
        if use_intermediate:                                         # :284-298
            z = gelu_erf(linear(y, params[pre + "intermediate.dense.weight"], params[pre + "intermediate.dense.bias"]))
            z = linear(z, params[pre + "output.dense.weight"], params[pre + "output.dense.bias"])
            z = _apply_keep(z, dm.get("ffn_out"), drop_p)            # :264
            y = layer_norm(z + y, params[pre + "output.LayerNorm.weight"], params[pre + "output.LayerNorm.bias"])
        x = y
    hidden = x
    att = probs.sum(dim=-3)                                          # :41  sum over heads  [N,M,M]
    comb = att.sum(dim=-2) / math.sqrt(M * H)                        # :42  sum over queries [N,M]
    w = torch.softmax(comb, dim=-1)                                  # :43
    joint = torch.cat([w[:, m:m + 1] * nrm(embs[m]) for m in range(M)], dim=1)       # :44-45
    w_fz = torch.softmax(params["weight_raw"], dim=0)                # :46  over all 6 entries
    joint_fz = torch.cat([w_fz[m] * nrm(embs[m]) for m in range(M)], dim=1)          # :48-49
    return joint, joint_fz, hidden, w


JOINT_WAYS = ("Mformer_hd_mean", "Mformer_hd_graph", "Mformer_weight", "atten_weight", "learnable_weight")


def atten_weight_readout(tokens: Tensor, a: Tensor) -> Tensor:
    """README joint_way "atten_weight" (README.md:51): non-transformer attention over the modality tokens [N, M, D].

    reference: the only arithmetic of that name in the tree, SelfAttention.forward model/EVA_tools.py:203-232 (dead code there):
    weights = inputs @ att_weights^T (:215-222), attentions = softmax(relu(weights)) over the tokens (:224), representations =
    sum_m attentions_m * inputs_m (:227-231).  PARITY UNPINNED beyond that formula: the MKGC source that wires it is absent.
    """
    pre = tokens @ a                                               # [N, M]
    w = torch.softmax(torch.relu(pre), dim=1)
    return (tokens * w.unsqueeze(-1)).sum(dim=1)


def adaptive_avg_pool_rows(x: Tensor, out_hw, view_hw=None) -> Tensor:
    """README POOL=1 / ``use_pool`` (README.md:68,79; no code in the tree, PARITY UNPINNED): each row of x [N, F] viewed as (H, W) and
    adaptively average-pooled to (oh, ow); (1, F) -> (1, P) is the 1-D pool.  torch's own definition of the windows is the specification."""
    N, Fw = x.shape
    H, W = (1, Fw) if view_hw is None else view_hw
    oh, ow = (1, out_hw) if isinstance(out_hw, int) else out_hw
    return F.adaptive_avg_pool2d(x.reshape(N, 1, H, W), (oh, ow)).reshape(N, oh * ow)


def readout(joint_way: str, joint: Tensor, joint_fz: Tensor, hidden: Tensor, graph_index: int) -> Tensor:
    """MKGC ``joint_way`` read-outs (README.md:51, names only; PARITY UNPINNED for hd_*).

    hd_mean = hidden.mean(1)   (cf. comment at model/MEAformer_tools.py:66-68)
    hd_graph = hidden[:, g]    (g = position of the structural token)
    Mformer_weight = joint ; learnable_weight = joint_fz (model/SNAG_tools.py:44-49)
    """
    if joint_way == "Mformer_hd_mean":
        return hidden.mean(dim=1)
    if joint_way == "Mformer_hd_graph":
        return hidden[:, graph_index, :]
    if joint_way == "Mformer_weight":
        return joint
    if joint_way == "learnable_weight":
        return joint_fz
    if joint_way == "atten_weight":
        raise ValueError("atten_weight reads the tokens, not the Mformer's outputs: atten_weight_readout(tokens, a)")
    raise ValueError(f"unknown joint_way {joint_way!r}")


# --------------------------------------------------------------------------
# a6: Gaussian noise-mask injection
# --------------------------------------------------------------------------


def col_mean_std(x: Tensor, exclude_rows: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Per-column mean and *unbiased* std over rows; optionally skip rows.

    reference: model/SNAG.py:77-84 (img stats skip ``ent_wo_img``), 94-95.
    """
    if exclude_rows is not None and exclude_rows.numel() > 0:
        keep = torch.ones(x.shape[0], dtype=torch.bool)
        keep[exclude_rows.long()] = False
        x = x[keep]
    return x.mean(dim=0), x.std(dim=0)


def add_noise_rows(x: Tensor, mean: Tensor, std: Tensor, row_mask: Tensor, z: Tensor, mask_ratio: float) -> Tensor:
    """X[mask] = (1-eps) X[mask] + eps (mean + std * z), z compact [n_masked, F].

    reference: model/SNAG.py:66-75 (mask = rand(N) < noise_ratio ; z = randn_like(X[mask])).
    Returns a new tensor (the reference mutates a clone, :89-91).
    """
    out = x.clone()
    sel = out[row_mask]
    noise = mean + std * z
    out[row_mask] = (1.0 - mask_ratio) * sel + mask_ratio * noise
    return out


def entity_noise_blend(e: Tensor, noise: Tensor, row_mask: Tensor, mask_ratio: float) -> Tensor:
    """e[mask] = (1 - eps/2) e[mask] + (eps/2) noise[mask]   (noise is a constant).

    reference: model/SNAG_tools.py:127-128 ; noise/mask built at model/SNAG.py:94-98.
    """
    m = row_mask.to(e.dtype).unsqueeze(1)
    a = mask_ratio * 0.5
    return e * (1.0 - a * m) + (a * m) * noise


# --------------------------------------------------------------------------
# a7: negative-sampled margin score
# --------------------------------------------------------------------------


def transe_l2sq(e: Tensor, r: Tensor, h_idx: Tensor, r_idx: Tensor, t_idx: Tensor) -> Tensor:
    """s(h,r,t) = || nrm(E[h]) + nrm(R[r]) - nrm(E[t]) ||_2^2.

    reference: model/MSNEA.py:110-121,151-152 (lower = more plausible).
    """
    dis = nrm(e[h_idx.long()]) + nrm(r[r_idx.long()]) - nrm(e[t_idx.long()])
    return dis.pow(2).sum(dim=-1)


def kge_intree(e: Tensor, r: Tensor, p_h, p_r, p_t, n_h, n_r, n_t, margin: float):
    """In-tree loss, term 1: sum(relu(margin + pos - neg)) with equal-length pos/neg lists.

    reference: model/MSNEA.py:120-122 (neg_triple_num == 1 in-tree).
    """
    pos = transe_l2sq(e, r, p_h, p_r, p_t)
    neg = transe_l2sq(e, r, n_h, n_r, n_t)
    return torch.relu(margin + pos - neg).sum(), pos, neg


def score_margin(e: Tensor, r: Tensor, h: Tensor, rel: Tensor, t: Tensor, neg_t: Tensor, margin: float):
    """MKGC shape: one positive (h,r,t) against K negative tails neg_t[b, :].

    loss = sum_b sum_k relu(margin + s(h,r,t) - s(h,r,neg_t[b,k]))  -- the K>1
    pairing is PARITY UNPINNED (north_star text); the score itself follows
    model/MSNEA.py:110-122.  Returns (loss, pos[B], neg[B,K]).
    """
    B, K = neg_t.shape
    q = nrm(e[h.long()]) + nrm(r[rel.long()])                      # [B,D]
    pos = (q - nrm(e[t.long()])).pow(2).sum(-1)                    # [B]
    tn = nrm(e[neg_t.long().reshape(-1)]).view(B, K, e.shape[1])
    neg = (q.unsqueeze(1) - tn).pow(2).sum(-1)                     # [B,K]
    loss = torch.relu(margin + pos.unsqueeze(1) - neg).sum()
    return loss, pos, neg


def selfadv_loss(e: Tensor, r: Tensor, h: Tensor, rel: Tensor, t: Tensor, neg_t: Tensor, margin: float, adv_temp: float):
    """Self-adversarial negative-sampling loss over the same scores (optional epilogue of SURVEY.md par. 8 a7).

    PARITY UNPINNED: the reference tree only carries the unused flags --adv_temp / --emb_dim (config.py:53-54); this restates the
    published formula with distances (lower = more plausible): p = softmax(-adv_temp * neg).detach();
    loss = sum_b [-logsigmoid(margin - pos) - sum_k p_k logsigmoid(neg_k - margin)].  Returns (loss, pos, neg).
    """
    _, pos, neg = score_margin(e, r, h, rel, t, neg_t, 0.0)
    p = torch.softmax(-adv_temp * neg, dim=1).detach()
    loss = (-torch.nn.functional.logsigmoid(margin - pos) - (p * torch.nn.functional.logsigmoid(neg - margin)).sum(1)).sum()
    return loss, pos, neg


def full_table_scores(e: Tensor, r: Tensor, h: Tensor, rel: Tensor) -> Tensor:
    """s(h, r, n) for every candidate tail n: [B, N] TransE-L2^2 scores on normalised rows (model/MSNEA.py:110-121,151-152), formed in
    chunks of 32 queries so that the [b, N, D] difference tensor stays small."""
    q = nrm(e[h.long()]) + nrm(r[rel.long()])
    en = nrm(e)
    out = [(q[b0:b0 + 32].unsqueeze(1) - en.unsqueeze(0)).pow(2).sum(-1) for b0 in range(0, q.shape[0], 32)]
    return torch.cat(out) if out else torch.zeros(0, e.shape[0], dtype=e.dtype)


def rank_full_table(e: Tensor, r: Tensor, h: Tensor, rel: Tensor, t: Tensor,
                    filter_mask: Optional[Tensor] = None) -> Tensor:
    """Link-prediction rank of the true tail against all N entities (count-less-than).

    PARITY UNPINNED (MKGC eval absent from the tree).  rank[b] = #{n : s(h,r,n) < s(h,r,t_b),
    n not filtered}.  ``filter_mask`` [B,N] bool marks known true tails to skip.
    """
    s = full_table_scores(e, r, h, rel)
    st = s.gather(1, t.long().view(-1, 1))
    less = s < st
    if filter_mask is not None:
        less = less & ~filter_mask.bool()
    return less.sum(dim=1)


# --------------------------------------------------------------------------
# a8: similarity evaluation
# --------------------------------------------------------------------------


def pairwise_distances(x: Tensor, y: Optional[Tensor] = None) -> Tensor:
    """clamp(|x|^2 + |y|^2 - 2 x y^T, 0).  reference: src/utils.py:202-218"""
    xn = x.pow(2).sum(1).view(-1, 1)
    if y is None:
        y = x
        yn = xn.view(1, -1)
    else:
        yn = y.pow(2).sum(1).view(1, -1)
    return torch.clamp(xn + yn - 2.0 * (x @ y.t()), min=0.0)


def csls_sim(sim: Tensor, k: int) -> Tensor:
    """C[i,j] = 2 S[i,j] - mean(topk_j S[i,:]) - mean(topk_i S[:,j]).  reference: src/utils.py:417-435"""
    r1 = torch.topk(sim, k, dim=1)[0].mean(1)
    r2 = torch.topk(sim.t(), k, dim=1)[0].mean(1)
    c = 2 * sim.t() - r1
    return c.t() - r2


def rank_metrics(emb_left: Tensor, emb_right: Tensor, csls_k: int = 0, top_k=(1, 10, 50)) -> dict:
    """Runner._test from the already-gathered, already-normalised rows.

    reference: main.py:376-436.  rank = #{j : Dist'[i,j] < Dist'[i,i]} which equals the
    index of i in the ascending sort when the row is tie-free.
    """
    dist = pairwise_distances(emb_left, emb_right)
    if csls_k and csls_k > 0:
        dist = 1 - csls_sim(1 - dist, csls_k)
    diag = dist.diagonal()
    rank_l2r = (dist < diag.view(-1, 1)).sum(dim=1)
    rank_r2l = (dist < diag.view(1, -1)).sum(dim=0)
    out = {"rank_l2r": rank_l2r, "rank_r2l": rank_r2l, "dist": dist}
    for name, rk in (("l2r", rank_l2r), ("r2l", rank_r2l)):
        n = rk.numel()
        rkf = rk.to(torch.float64)
        out[f"hits_{name}"] = [round(float((rk < k).sum().item()) / n, 4) for k in top_k]
        out[f"mr_{name}"] = float((rkf + 1).mean())
        out[f"mrr_{name}"] = float((1.0 / (rkf + 1)).mean())
    return out


def mutual_nearest(final_emb: Tensor, left: Tensor, right: Tensor, chunk: int = 1000):
    """Iter_new_links' distance part: row argmin / column argmin / mutual pairs.

    reference: model/SNAG.py:195-204 (1000-row chunks, first-branch filter).
    """
    ds = [pairwise_distances(final_emb[left[i:i + chunk]], final_emb[right]) for i in range(0, len(left), chunk)]
    dist = torch.cat(ds, dim=0)
    pl = torch.argmin(dist, dim=1)
    pr = torch.argmin(dist.t(), dim=1)
    pairs = [(int(left[i]), int(right[p])) for i, p in enumerate(pl.tolist()) if int(pr[p]) == i]
    return pl, pr, pairs


# --------------------------------------------------------------------------
# (f) next #1: MMEA contrastive objective
# --------------------------------------------------------------------------


def icl_loss(emb: Tensor, links: Tensor, tau: float, ab_weight: float = 0.5, weight_norm: Optional[Tensor] = None,
             norm: bool = True) -> Tensor:
    """Bidirectional in-batch InfoNCE over aligned pairs.  reference: icl_loss.forward model/SNAG_loss.py:58-128
    (neg_l / neg_r / inversion unused by SNAG: model/SNAG.py:50-51,106-112,143-162).

    z1 = nrm(emb)[links[:,0]], z2 = nrm(emb)[links[:,1]];  logits_a = [z1 z2^T | z1 z1^T - 1e9 I] / tau, logits_b likewise
    with the roles swapped; loss = alpha * CE_a + (1 - alpha) * CE_b, CE = -mean_i w_i log_softmax(logits)[i, i],
    w_i = min(weight_norm[l_i], weight_norm[r_i]) when weight_norm is given (:66-70, softXEnt :42-56).
    """
    if norm:
        emb = nrm(emb)
    li, ri = links[:, 0].long(), links[:, 1].long()
    z1, z2 = emb[li], emb[ri]
    B = z1.shape[0]
    eye = torch.eye(B, dtype=emb.dtype)
    la = torch.cat([z1 @ z2.t() / tau, z1 @ z1.t() / tau - eye * 1e9], dim=1)
    lb = torch.cat([z2 @ z1.t() / tau, z2 @ z2.t() / tau - eye * 1e9], dim=1)
    w = torch.minimum(weight_norm[li], weight_norm[ri]) if weight_norm is not None else torch.ones(B, dtype=emb.dtype)
    idx = torch.arange(B)
    ce_a = -(torch.log_softmax(la, dim=1)[idx, idx] * w).sum() / B
    ce_b = -(torch.log_softmax(lb, dim=1)[idx, idx] * w).sum() / B
    return ab_weight * ce_a + (1.0 - ab_weight) * ce_b


def multi_loss(losses: Sequence[Tensor], log_vars: Tensor) -> Tensor:
    """CustomMultiLossLayer: sum_i exp(-s_i) L_i + s_i.  reference: model/SNAG_loss.py:12-29"""
    tot = 0
    for i, l in enumerate(losses):
        tot = tot + torch.exp(-log_vars[i]) * l + log_vars[i]
    return tot


# --------------------------------------------------------------------------
# (f) next #2: structural encoder (sparse GAT, diag variant)
# --------------------------------------------------------------------------


def gat_forward(x: Tensor, edges: Tensor, layers: Sequence[Dict[str, Tensor]],
                in_keep: Optional[Sequence[Optional[Tensor]]] = None, edge_keep: Optional[Sequence[Optional[Tensor]]] = None,
                p_in: float = 0.0, p_edge: float = 0.0) -> Tensor:
    """GAT(diag=True): eval mode by default (the SNAG configuration: run_snag.sh sets neither --dropout nor --attn_dropout); training mode with
    explicit keep masks: ``in_keep[l]`` [N, f] = F.dropout on the input of layer l (every layer but the last, model/Tool_model.py:78-79),
    ``edge_keep[l]`` [H, E] = F.dropout on head i's edge weights AFTER the row sums (model/layers.py:81-83), inverted dropout at rates p_in / p_edge.

    reference: MultiHeadGraphAttention.forward model/layers.py:68-94 and GAT.forward model/Tool_model.py:75-94.
    Per layer and head i: h = x * w_i ; e = exp(-leaky_relu_0.2([h_u | h_v] . a_i)) per edge (u, v) ; h'_u = sum_e e h_v / sum_e e ;
    heads are averaged; ELU between layers.  ``edges`` is the [2, E] index list of the COO adjacency as stored (duplicates and
    self loops included, values unused); ``layers[l]`` holds ``w`` [H,1,f] and ``a_src_dst`` [H,2f,1].
    """
    src, dst = edges[0].long(), edges[1].long()
    N = x.shape[0]
    for li, p in enumerate(layers):
        w, a = p["w"], p["a_src_dst"]
        H, f = w.shape[0], w.shape[2]
        if in_keep is not None and in_keep[li] is not None:
            x = x * in_keep[li].to(x.dtype) / (1.0 - p_in)                   # Tool_model.py:78-79
        outs = []
        for i in range(H):
            h = x * w[i]                                                     # :74
            z = torch.cat([h[src], h[dst]], dim=1) @ a[i]                    # :78-79
            e = torch.exp(-F.leaky_relu(z.squeeze(1), 0.2))                  # :79
            rowsum = torch.zeros(N, dtype=x.dtype).index_add(0, src, e)      # :81 (spmm with ones)
            if edge_keep is not None and edge_keep[li] is not None:
                e = e * edge_keep[li][i].to(x.dtype) / (1.0 - p_edge)        # :83
            hp = torch.zeros(N, f, dtype=x.dtype).index_add(0, src, e.unsqueeze(1) * h[dst])      # :85
            outs.append(hp / rowsum.unsqueeze(1))                            # :86
        x = torch.stack(outs, 0).mean(dim=0)                                 # Tool_model.py:84-85
        if li + 1 < len(layers):
            x = F.elu(x)                                                     # :87-88
    return x


# --------------------------------------------------------------------------
# whole MKGC step (fusion + score), used by the CPU baseline and end-to-end tests
# --------------------------------------------------------------------------


def gcn_forward(x: Tensor, adj: Tensor, p: Dict[str, Tensor]) -> Tensor:
    """Two-layer GCN, dropout 0: gc2(relu(gc1(x, adj)), adj) with gc(x, adj) = spmm(adj, x @ W) + b.

    reference: GraphConvolution.forward model/layers.py:124-130, GCN.forward model/Tool_model.py:106-110.  ``p`` uses the reference's
    state_dict keys (gc1.weight [in, out], gc1.bias, gc2.weight, gc2.bias); ``adj`` is a sparse COO tensor (values used).
    """
    h = torch.sparse.mm(adj, x @ p["gc1.weight"]) + p["gc1.bias"]
    h = torch.relu(h)
    return torch.sparse.mm(adj, h @ p["gc2.weight"]) + p["gc2.bias"]


def mkgc_forward(feats: Sequence[Tensor], proj: Sequence[Dict[str, Tensor]], ent_table: Tensor,
                 fusion_params: Dict[str, Tensor], rel_table: Tensor,
                 h: Tensor, rel: Tensor, t: Tensor, neg_t: Tensor, *,
                 num_heads: int, margin: float, joint_way: str = "Mformer_hd_graph",
                 use_intermediate: bool = False, num_layers: int = 1,
                 ent_noise: Optional[Tensor] = None, ent_noise_mask: Optional[Tensor] = None,
                 mask_ratio: float = 0.0, loss_kind: str = "hinge", adv_temp: float = 1.0,
                 drop_masks: Optional[List[Dict[str, Tensor]]] = None, drop_p: float = 0.1,
                 proj_keep: Optional[Sequence[Optional[Tensor]]] = None, proj_p: float = 0.0,
                 atten_a: Optional[Tensor] = None):
    """One MKGC hot-path step: projections -> tokens -> Mformer -> read-out -> margin score.

    Token order is [modality_0, ..., modality_{P-1}, structural]; the structural token is the
    (optionally noise-blended) entity-table row (model/SNAG_tools.py:122-152 with the GAT removed,
    which is how SURVEY par. 8 a5 scopes the MKGC structural token).  ``proj[i]`` holds either
    {"weight","bias"} (num_proj=1) or {"l1","l2"} (+ optional "weight","bias" applied first).

    Training mode with explicit masks: ``drop_masks`` / ``drop_p`` are ``fusion_forward``'s (the Bert dropouts,
    model/SNAG_tools.py:201,220,264); ``proj_keep[i]`` / ``proj_p`` the 0/1 keep mask [N, D] and rate of modality i's
    ProjectionHead dropout (model/layers.py:142-147).
    """
    toks = []
    for i, (f, p) in enumerate(zip(feats, proj)):
        if "weight" in p:
            f = linear(f, p["weight"], p.get("bias"))
        if "l1" in p:
            f = projection_head(f, p["l1"], p["l2"], keep_mask=proj_keep[i] if proj_keep is not None else None, p=proj_p)
        toks.append(f)
    e = ent_table
    if ent_noise is not None:
        e = entity_noise_blend(e, ent_noise, ent_noise_mask, mask_ratio)
    toks.append(e)
    if joint_way == "atten_weight":          # no transformer: the read-out is a function of the tokens and ``atten_a`` [D] alone
        ent = atten_weight_readout(torch.stack(toks, dim=1), atten_a)
    else:
        joint, joint_fz, hidden, w = fusion_forward(toks, fusion_params, num_heads, num_layers, use_intermediate,
                                                    drop_masks=drop_masks, drop_p=drop_p)
        ent = readout(joint_way, joint, joint_fz, hidden, graph_index=len(toks) - 1)
    if loss_kind == "self_adversarial":
        loss, pos, neg = selfadv_loss(ent, rel_table, h, rel, t, neg_t, margin, adv_temp)
    else:
        loss, pos, neg = score_margin(ent, rel_table, h, rel, t, neg_t, margin)
    return loss, pos, neg, ent
