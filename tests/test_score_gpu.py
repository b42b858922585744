"""GPU parity: snag_score_margin_fwd/bwd (through the C ABI) vs the CPU oracle and golden G3."""
import numpy as np
import pytest
import torch

from helpers import T, assert_close, load
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(N, R, D, B, K, seed, idt=torch.int64, scale=0.5):
    g = torch.Generator().manual_seed(seed)
    ent = torch.randn(N, D, generator=g) * scale
    rel = torch.randn(R, D, generator=g) * 0.3
    h = torch.randint(0, N, (B,), generator=g).to(idt)
    r = torch.randint(0, R, (B,), generator=g).to(idt)
    t = torch.randint(0, N, (B,), generator=g).to(idt)
    neg = torch.randint(0, N, (B, K), generator=g).to(idt)
    return ent, rel, h, r, t, neg


@pytest.mark.parametrize("D,K,B,idt", [
    (128, 32, 1024, torch.int32), (128, 32, 257, torch.int64), (128, 1, 64, torch.int64), (128, 0, 33, torch.int32),
    (128, 3, 50, torch.int32), (128, 33, 40, torch.int64), (128, 70, 19, torch.int32), (128, 128, 16, torch.int32),
    (64, 32, 100, torch.int32), (256, 128, 48, torch.int32), (300, 7, 30, torch.int64), (100, 5, 21, torch.int32),
    (512, 9, 12, torch.int32), (4, 2, 9, torch.int32),
    # several waves per triple (two at 96 <= K < 128, four from K = 128), ragged shares, a dead wave at the end of the grid
    (128, 100, 37, torch.int32), (256, 130, 9, torch.int64), (128, 129, 3, torch.int32), (64, 97, 2, torch.int32),
])
def test_score_fwd_bwd_vs_oracle(D, K, B, idt):
    from snag_amd import ops
    margin = 1.2
    ent, rel, h, r, t, neg = _mk(501, 13, D, B, K, seed=D * 1000 + K, idt=idt)
    ent[7].zero_()                                   # normalize eps branch
    h[0] = 7
    if K:
        neg[1, 0] = 7
    e_ = ent.clone().requires_grad_(True)
    r_ = rel.clone().requires_grad_(True)
    loss, pos, ng = O.score_margin(e_, r_, h, r, t, neg, margin)
    # keep the eps-branch row out of the gradient comparison scale (1e12 magnitudes) by weighting
    loss.backward()
    d = [x.to(DEV) for x in (ent, rel, h, r, t, neg)]
    pos_g, neg_g, lp = ops.score_margin_fwd(*d, margin)
    assert_close(pos_g, pos, atol=1e-5, what="pos")
    assert_close(neg_g, ng, atol=1e-5, what="neg")
    assert_close(ops.reduce_sum(lp), loss, atol=1e-4, rtol=1e-5, what="loss")
    assert_close(lp, torch.relu(margin + pos.unsqueeze(1) - ng).sum(1), atol=1e-4, rtol=2e-6, what="loss_part")
    de, dr = ops.score_margin_bwd(*d, margin)
    assert_close(de, e_.grad, atol=2e-4, rtol=1e-4, what="d_ent")
    assert_close(dr, r_.grad, atol=2e-4, rtol=1e-4, what="d_rel")


_FUSED_SHAPES = [
    (128, 32, 1024, torch.int32), (128, 32, 257, torch.int64), (128, 1, 64, torch.int64), (128, 0, 33, torch.int32),
    (128, 3, 50, torch.int32), (128, 33, 40, torch.int64), (128, 70, 19, torch.int32), (128, 128, 16, torch.int32),
    (64, 32, 100, torch.int32), (256, 128, 48, torch.int32), (300, 7, 30, torch.int64), (100, 5, 21, torch.int32),
    (512, 9, 12, torch.int32), (4, 2, 9, torch.int32), (128, 100, 37, torch.int32), (256, 130, 9, torch.int64),
    (128, 129, 3, torch.int32), (64, 97, 2, torch.int32), (128, 16, 5, torch.int32), (128, 17, 6, torch.int32),
]


@pytest.mark.parametrize("D,K,B,idt", _FUSED_SHAPES)
@pytest.mark.parametrize("margin", [1.2, 12.0])
def test_score_fused_entry_vs_oracle(D, K, B, idt, margin):
    """snag_score_margin_fwd_bwd (one launch: scores + per-triple losses + summed loss + scattered gradients) on the same cases as
    the two-launch form.  margin 1.2 leaves part of the hinges inactive (their negatives must get NO gradient); margin 12 is the
    README value, every hinge active (scores of unit rows are <= 9)."""
    from snag_amd import ops
    ent, rel, h, r, t, neg = _mk(501, 13, D, B, K, seed=D * 1000 + K, idt=idt)
    ent[7].zero_()                                   # normalize eps branch
    h[0] = 7
    if K:
        neg[1, 0] = 7
    e_ = ent.clone().requires_grad_(True)
    r_ = rel.clone().requires_grad_(True)
    loss, pos, ng = O.score_margin(e_, r_, h, r, t, neg, margin)
    (0.5 * loss).backward()
    d = [x.to(DEV) for x in (ent, rel, h, r, t, neg)]
    lg, pos_g, neg_g, lp, de, dr = ops.score_margin_fwd_bwd(*d, margin, grad_scale=0.5)
    assert_close(pos_g, pos, atol=1e-5, what="pos")
    assert_close(neg_g, ng, atol=1e-5, what="neg")
    assert_close(lp, torch.relu(margin + pos.unsqueeze(1) - ng).sum(1), atol=1e-4, rtol=2e-6, what="loss_part")
    assert_close(lg, loss, atol=1e-4, rtol=1e-5, what="loss")
    # the in-kernel sum is the fixed-order sum of loss_part: identical to the separate reduction kernel over the same values
    assert_close(lg, lp.double().sum(), atol=0, rtol=2e-6, what="loss == sum(loss_part)")
    assert_close(de, e_.grad, atol=2e-4, rtol=1e-4, what="d_ent")
    assert_close(dr, r_.grad, atol=2e-4, rtol=1e-4, what="d_rel")
    # same scores as the forward-only kernel (different lane layout and summation order: equal to rounding)
    pos_f, neg_f, _ = ops.score_margin_fwd(*d, margin)
    assert_close(pos_g, pos_f, atol=3e-6)
    assert_close(neg_g, neg_f, atol=3e-6)


@pytest.mark.parametrize("D,K,B,idt", [c for c in _FUSED_SHAPES if c[0] % 4 == 0])
@pytest.mark.parametrize("margin", [1.2, 12.0])
def test_score_segmented_entry_vs_oracle(D, K, B, idt, margin):
    """snag_score_margin_fwd_bwd_seg: no float atomics (per-triple pass + per-destination fixed-point sums).  Same cases as the
    atomic form; in addition the outputs of two launches are identical BITS, and the gradient buffers are overwritten (stale
    contents must not leak through)."""
    from snag_amd import ops
    ent, rel, h, r, t, neg = _mk(501, 13, D, B, K, seed=D * 1000 + K, idt=idt)
    ent[7].zero_()                                   # normalize eps branch
    h[0] = 7
    if K:
        neg[1, 0] = 7
        if B > 2:
            neg[2, :] = 3                            # one entity drawn K times by one triple, and ...
        t[3:3 + max(0, min(B - 3, 40))] = 5          # ... a hub tail: more entries than a destination's bucket holds (overflow list)
    e_ = ent.clone().requires_grad_(True)
    r_ = rel.clone().requires_grad_(True)
    loss, pos, ng = O.score_margin(e_, r_, h, r, t, neg, margin)
    (0.5 * loss).backward()
    d = [x.to(DEV) for x in (ent, rel, h, r, t, neg)]
    stale_e, stale_r = torch.full((501, D), 7.0, device=DEV), torch.full((13, D), -3.0, device=DEV)
    lg, pos_g, neg_g, lp, de, dr = ops.score_margin_fwd_bwd_seg(*d, margin, grad_scale=0.5, d_ent=stale_e, d_rel=stale_r)
    assert_close(pos_g, pos, atol=1e-5, what="pos")
    assert_close(neg_g, ng, atol=1e-5, what="neg")
    assert_close(lg, loss, atol=1e-4, rtol=1e-5, what="loss")
    assert_close(de, e_.grad, atol=2e-4, rtol=1e-4, what="d_ent")
    assert_close(dr, r_.grad, atol=2e-4, rtol=1e-4, what="d_rel")
    pos_f, neg_f, _ = ops.score_margin_fwd(*d, margin)
    assert torch.equal(pos_g, pos_f) and torch.equal(neg_g, neg_f)            # pass A is the forward kernel's arithmetic
    lg2, _, _, lp2, de2, dr2 = ops.score_margin_fwd_bwd_seg(*d, margin, grad_scale=0.5)
    assert torch.equal(de2, de) and torch.equal(dr2, dr) and torch.equal(lp2, lp) and torch.equal(lg2, lg)
    # sparse-write form: rows nothing arrived at keep what they held
    keep_e = torch.full((501, D), 9.0, device=DEV)
    _, _, _, _, de3, _ = ops.score_margin_fwd_bwd_seg(*d, margin, grad_scale=0.5, d_ent=keep_e, d_rel=torch.zeros(13, D, device=DEV), write_untouched=False)
    touched = torch.zeros(501, dtype=torch.bool)                 # every head, tail and negative tail of the batch (active hinge or not)
    touched[h.long()] = True
    touched[t.long()] = True
    touched[neg.flatten().long()] = True
    assert torch.equal(de3[touched.to(DEV)], de[touched.to(DEV)]) and float((de3[~touched.to(DEV)] - 9.0).abs().max() if (~touched).any() else 0.0) == 0.0


@pytest.mark.parametrize("D,K,B,idt", [(128, 32, 100, torch.int32), (256, 128, 48, torch.int64), (64, 3, 50, torch.int32), (8, 2, 9, torch.int32),
                                        (512, 9, 12, torch.int32), (104, 5, 21, torch.int32), (128, 32, 700, torch.int32)])
@pytest.mark.parametrize("tab", ["f32", "bf16"])
def test_score_touched_form_vs_oracle_and_the_segmented_form(D, K, B, idt, tab):
    """snag_score_margin_fwd_bwd_touched(_bf16): the segmented scorer whose second pass walks the batch's entries instead of every
    destination.  Against the oracle (bf16 tables: fp32 arithmetic on the bf16-rounded values); for fp32 tables scores and losses
    bit-equal to the all-destinations form and, while a triple is one part (K < 48), the gradients too; hot rows (one entity drawn by
    hundreds of entries: inline bucket + several hash chunks + the hot-row pass) included; three launches in a row on one workspace give
    identical bits (counters, hash keys and hot-row list are handed over clean)."""
    from snag_amd import ops
    N, R, margin = 2001, 13, 6.0
    ent, rel, h, r, t, neg = _mk(N, R, D, B, K, seed=D * 77 + K, idt=idt)
    ent[7].zero_()
    h[0] = 7
    if B > 40:
        t[3:3 + B // 2] = 5                              # a hub tail ...
        neg[B // 3:, 0] = 11                             # ... and a hot negative: > 16 + 64 entries at B = 700 (two hash chunks)
        neg[2, :] = 3
    if tab == "bf16":
        ent, rel = ent.bfloat16(), rel.bfloat16()
    e_ = ent.float().clone().requires_grad_(True)
    r_ = rel.float().clone().requires_grad_(True)
    loss, pos, ng = O.score_margin(e_, r_, h, r, t, neg, margin)
    (0.5 * loss).backward()
    d = [x.to(DEV) for x in (ent, rel, h, r, t, neg)]
    outs = []
    for _ in range(3):
        outs.append(ops.score_margin_fwd_bwd_touched(*d, margin, grad_scale=0.5))
    lg, pos_g, neg_g, lp, de, dr = outs[0]
    assert_close(pos_g, pos, atol=1e-5, what="pos")
    assert_close(neg_g, ng, atol=1e-5, what="neg")
    assert_close(lg, loss, atol=1e-4, rtol=1e-5, what="loss")
    assert_close(de, e_.grad, atol=2e-4, rtol=1e-4, what="d_ent")
    assert_close(dr, r_.grad, atol=2e-4, rtol=1e-4, what="d_rel")
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
    if tab == "f32":
        ref = ops.score_margin_fwd_bwd_seg(*d, margin, grad_scale=0.5)
        assert torch.equal(ref[1], pos_g) and torch.equal(ref[2], neg_g)
        touched = torch.zeros(N, dtype=torch.bool)
        touched[torch.cat([h, t, neg.flatten()]).long()] = True
        assert float(de[~touched.to(DEV)].abs().sum()) == 0.0
        rels = torch.unique(r.long()).to(DEV)
        if K < 48:                                       # one wave per triple on both sides: the same sums in the same order
            assert torch.equal(ref[0], lg) and torch.equal(ref[4][touched.to(DEV)], de[touched.to(DEV)]) and torch.equal(ref[5][rels], dr[rels])
        else:                                            # the query gradient is dealt to several parts: equal to fp32 rounding
            assert_close(lg, ref[0], atol=1e-4, rtol=1e-6)
            assert_close(de, torch.where(touched.to(DEV)[:, None], ref[4], torch.zeros_like(ref[4])), atol=1e-5, rtol=1e-5)


def test_score_fused_accumulates_into_strided_rows_and_resets_its_ticket():
    """The engine's use: the table is the slot hidden[:, g, :] of a stacked [N, M, D] tensor (row stride M*D) and the gradient is
    scatter-added into the same slot of d_hidden; two launches in a row accumulate; the ticket word is zero again after each."""
    from snag_amd import ops
    from snag_amd._lib import call, ptr, stream
    N, R, D, B, K, M = 400, 9, 128, 300, 32, 3
    ent, rel, h, r, t, neg = _mk(N, R, D, B, K, seed=77, idt=torch.int32)
    hidden = torch.randn(N, M, D, generator=torch.Generator().manual_seed(1))
    hidden[:, 2, :] = ent
    hd = hidden.to(DEV)
    table = hd[:, 2, :]
    d_hidden = torch.zeros(N, M, D, device=DEV)
    d_rel = torch.zeros(R, D, device=DEV)
    e_ = ent.clone().requires_grad_(True)
    r_ = rel.clone().requires_grad_(True)
    loss, _, _ = O.score_margin(e_, r_, h, r, t, neg, 12.0)
    loss.backward()
    d = [x.to(DEV) for x in (rel, h, r, t, neg)]
    tick = ops._ticket(torch.device(DEV, torch.cuda.current_device()))
    for rep in (1, 2):
        lg, *_ = ops.score_margin_fwd_bwd(table, *d, 12.0, d_ent=d_hidden[:, 2, :], d_rel=d_rel)
        assert int(tick.abs().sum().item()) == 0
        assert_close(lg, loss, atol=1e-3, rtol=1e-5)
        assert_close(d_hidden[:, 2, :], rep * e_.grad, atol=4e-4, rtol=1e-4, what=f"d_hidden slot after {rep} launches")
        assert float(d_hidden[:, :2, :].abs().max()) == 0.0
        assert_close(d_rel, rep * r_.grad, atol=4e-4, rtol=1e-4)


def test_score_fused_loss_reduction_under_load():
    """The last-finishing block sums loss_part written by every other block (write-through stores + ticket).  Large uneven grids,
    many repeats, an L2-warming read of loss_part in between: the sum must equal the fixed-order sum of what is in memory."""
    from snag_amd import ops
    ent, rel, h, r, t, neg = [x.to(DEV) for x in _mk(15000, 558, 128, 20000 + 3, 32, seed=3, idt=torch.int32)]
    for rep in range(12):
        B = (20003, 1024, 4099, 257)[rep % 4]
        lg, _, _, lp, _, _ = ops.score_margin_fwd_bwd(ent, rel, h[:B], r[:B], t[:B], neg[:B], 12.0)
        ref = ops.reduce_sum(lp)
        assert abs(float(lg) - float(ref)) <= 2e-6 * abs(float(ref)), (rep, B, float(lg), float(ref))
        assert abs(float(lg) - float(lp.double().sum())) <= 1e-5 * abs(float(ref))


def test_margin_loss_autograd_runs_the_fused_launch():
    """snag_amd.score.margin_loss under autograd: one launch in forward, device-side scaling by the upstream gradient."""
    from snag_amd.score import margin_loss
    ent, rel, h, r, t, neg = _mk(300, 7, 128, 90, 32, seed=21)
    e_ = ent.clone().requires_grad_(True)
    r_ = rel.clone().requires_grad_(True)
    loss, _, _ = O.score_margin(e_, r_, h, r, t, neg, 12.0)
    (loss * 0.25).backward()
    eg, rg = ent.to(DEV).requires_grad_(True), rel.to(DEV).requires_grad_(True)
    lg, pg, ngg = margin_loss(eg, rg, h.to(DEV), r.to(DEV), t.to(DEV), neg.to(DEV), 12.0)
    (lg * 0.25).backward()
    assert_close(lg, loss, atol=1e-3, rtol=1e-5)
    assert_close(eg.grad, e_.grad, atol=1e-4, rtol=1e-4)
    assert_close(rg.grad, r_.grad, atol=1e-4, rtol=1e-4)
    with torch.no_grad():                                     # evaluation: forward kernel only
        l2, _, _ = margin_loss(eg, rg, h.to(DEV), r.to(DEV), t.to(DEV), neg.to(DEV), 12.0)
    assert_close(l2, loss, atol=1e-3, rtol=1e-5)


def test_score_bwd_explicit_mode():
    from snag_amd import ops
    ent, rel, h, r, t, neg = _mk(300, 7, 128, 77, 5, seed=5)
    g = torch.Generator().manual_seed(9)
    dpos, dneg = torch.randn(77, generator=g), torch.randn(77, 5, generator=g)
    e_ = ent.clone().requires_grad_(True)
    r_ = rel.clone().requires_grad_(True)
    _, pos, ng = O.score_margin(e_, r_, h, r, t, neg, 0.0)
    ((pos * dpos).sum() + (ng * dneg).sum()).backward()
    d = [x.to(DEV) for x in (ent, rel, h, r, t, neg)]
    de, dr = ops.score_margin_bwd(*d, 0.0, grad_scale=1.0, dpos=dpos.to(DEV), dneg=dneg.to(DEV))
    assert_close(de, e_.grad, atol=1e-4, rtol=1e-4)
    assert_close(dr, r_.grad, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("K,B,margin,temp", [(32, 257, 12.0, 1.0), (5, 33, 2.0, 0.5), (70, 19, 6.0, 2.0), (1, 9, 1.0, 1.0)])
def test_self_adversarial_loss_vs_oracle(K, B, margin, temp):
    """Optional a7 epilogue (parity unpinned: the oracle restates the published formula)."""
    from snag_amd.score import self_adversarial_loss
    ent, rel, h, r, t, neg = _mk(400, 11, 128, B, K, seed=K * 7 + B)
    e_ = ent.clone().requires_grad_(True)
    r_ = rel.clone().requires_grad_(True)
    loss, pos, ng = O.selfadv_loss(e_, r_, h, r, t, neg, margin, temp)
    loss.backward()
    eg = ent.to(DEV).requires_grad_(True)
    rg = rel.to(DEV).requires_grad_(True)
    lg, pg, ngg = self_adversarial_loss(eg, rg, h.to(DEV), r.to(DEV), t.to(DEV), neg.to(DEV), margin, temp)
    lg.backward()
    assert_close(pg, pos.detach(), atol=1e-5)
    assert_close(ngg, ng.detach(), atol=1e-5)
    assert_close(lg, loss.detach(), atol=1e-3, rtol=1e-5, what="loss")
    assert_close(eg.grad, e_.grad, atol=1e-4, rtol=1e-4, what="d_ent")
    assert_close(rg.grad, r_.grad, atol=1e-4, rtol=1e-4, what="d_rel")


def test_score_golden_intree_kge():
    """golden G3: MSNEA.kge term 1 (model/MSNEA.py:110-122), head-or-tail corruption, one negative per positive."""
    from snag_amd.score import kge_relation_loss
    z = load("g3_score.npz")
    ent = T(z["ent"], DEV).requires_grad_(True)
    rel = T(z["rel"], DEV).requires_grad_(True)
    idx = [T(z[k], DEV) for k in ("p_h", "p_r", "p_t", "n_h", "n_r", "n_t")]
    loss, pos, neg = kge_relation_loss(ent, rel, *idx, margin=float(z["margin"]))
    assert_close(pos, z["pos"], atol=1e-5)
    assert_close(neg, z["neg"], atol=1e-5)
    assert_close(loss, z["loss_term1"], atol=1e-4)
    loss.backward()
    assert_close(ent.grad, z["g.ent"], atol=1e-4, rtol=1e-4)
    assert_close(rel.grad, z["g.rel"], atol=1e-4, rtol=1e-4)


def test_score_full_size_properties():
    """BASELINE config 2 shape (|E|=15k, D=128, B=1024, K=32): size-independent properties."""
    from snag_amd import ops
    ent, rel, h, r, t, neg = [x.to(DEV) for x in _mk(15000, 558, 128, 1024, 32, seed=1234, idt=torch.int32)]
    pos, ng, lp = ops.score_margin_fwd(ent, rel, h, r, t, neg, 12.0)
    # scale invariance: rows are L2-normalised, so scaling the tables must not change any score
    pos2, ng2, _ = ops.score_margin_fwd(ent * 3.0, rel * 0.25, h, r, t, neg, 12.0)
    assert_close(pos2, pos, atol=2e-6)
    assert_close(ng2, ng, atol=2e-6)
    # a negative tail equal to the positive tail scores exactly like the positive
    neg2 = neg.clone()
    neg2[:, 5] = t
    _, ng3, _ = ops.score_margin_fwd(ent, rel, h, r, t, neg2, 12.0)
    assert torch.equal(ng3[:, 5], pos)
    # bounds: 0 <= s <= 9 for unit vectors; margin 12 keeps every hinge active -> loss = sum(12 + pos - neg)
    assert float(ng.min()) >= 0 and float(ng.max()) <= 9.0 + 1e-4
    assert_close(lp, (12.0 + pos.unsqueeze(1) - ng).sum(1), atol=2e-4)
    # permutation of negatives permutes scores
    perm = torch.randperm(32, device=DEV)
    _, ng4, _ = ops.score_margin_fwd(ent, rel, h, r, t, neg[:, perm].contiguous(), 12.0)
    assert torch.equal(ng4, ng[:, perm])
    # gradient: total mass conservation -- sum over all rows of d_ent . ent_row is 0 (normalize is scale invariant)
    de, dr = ops.score_margin_bwd(ent, rel, h, r, t, neg, 12.0)
    assert float((de * ent).sum(1).abs().max()) < 1e-3
    assert float((dr * rel).sum(1).abs().max()) < 1e-3


def test_score_empty_batch():
    from snag_amd import ops
    ent, rel, h, r, t, neg = [x.to(DEV) for x in _mk(10, 3, 128, 0, 4, seed=1)]
    pos, ng, lp = ops.score_margin_fwd(ent, rel, h, r, t, neg, 1.0)
    assert pos.numel() == 0 and ng.shape == (0, 4)


@pytest.mark.parametrize("D,K,B", [(256, 128, 64), (128, 32, 100), (64, 5, 33), (512, 17, 9), (128, 100, 33), (256, 130, 10), (128, 129, 3)])
def test_score_bf16_tables(D, K, B):
    """Config-5 storage: bf16 tables, fp32 math.  The oracle runs on the same bf16-rounded values, so parity stays 1e-5."""
    from snag_amd import ops
    ent, rel, h, r, t, neg = _mk(3000, 13, D, B, K, seed=D + K, idt=torch.int32)
    eb, rb = ent.to(torch.bfloat16), rel.to(torch.bfloat16)
    loss, pos, ng = O.score_margin(eb.float(), rb.float(), h, r, t, neg, 12.0)
    pos_g, neg_g, lp = ops.score_margin_fwd_bf16(eb.to(DEV), rb.to(DEV), h.to(DEV), r.to(DEV), t.to(DEV), neg.to(DEV), 12.0)
    assert_close(pos_g, pos, atol=1e-5)
    assert_close(neg_g, ng, atol=1e-5)
    assert_close(ops.reduce_sum(lp), loss, atol=1e-3, rtol=2e-6)


def test_device_negative_sampler():
    """Unpinned by the reference (RNG streams differ): check the sampler's contract instead."""
    from snag_amd import ops
    N, R, B, K = 2000, 7, 4096, 32
    g = torch.Generator().manual_seed(3)
    h = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(DEV)
    r = torch.randint(0, R, (B,), generator=g, dtype=torch.int32).to(DEV)
    t = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(DEV)
    neg = ops.sample_negatives(h, r, t, K, N, R, seed=11)
    assert neg.shape == (B, K) and int(neg.min()) >= 0 and int(neg.max()) < N
    assert torch.equal(neg, ops.sample_negatives(h, r, t, K, N, R, seed=11))          # counter-based: reproducible
    assert not torch.equal(neg, ops.sample_negatives(h, r, t, K, N, R, seed=12))
    counts = torch.bincount(neg.flatten().long(), minlength=N).float()
    assert abs(float(counts.mean()) - B * K / N) < 1e-3 and float(counts.std()) < 1.5 * (B * K / N) ** 0.5     # ~Poisson
    # filter: make EVERY (h, r, x) with x even a known triple for the first 64 positives -> their negatives must be odd
    hh, rr = h[:64].cpu(), r[:64].cpu()
    ev = torch.arange(0, N, 2)
    kh = hh.repeat_interleave(ev.numel()); kr = rr.repeat_interleave(ev.numel()); kt = ev.repeat(64)
    keys = ops.triple_keys(kh.to(DEV), kr.to(DEV), kt.to(DEV), N, R)
    negf = ops.sample_negatives(h[:64], r[:64], t[:64], K, N, R, seed=5, known_sorted=keys)
    frac_even = float((negf % 2 == 0).float().mean())
    assert frac_even < 0.01, frac_even                     # only the "10th try is kept" escape (p = 2^-10) may leak
    # in-tree mode: per positive either all heads or all tails are corrupted, about half each
    nh, nt = ops.sample_negatives(h, r, t, K, N, R, seed=9, head_or_tail=True)
    head_rows = (nt == t.view(-1, 1)).all(1)
    tail_rows = (nh == h.view(-1, 1)).all(1)
    assert bool((head_rows | tail_rows).all())
    assert 0.45 < float(head_rows.float().mean()) < 0.55


def test_score_bwd_fixed_point_is_reproducible_and_equal_to_the_float_path():
    """snag_score_margin_bwd_fixed + snag_fixed_to_float: integer atomics, so heavy collisions (every negative drawn from 8 rows)
    give identical bits run after run; values equal the fp32-atomic kernel's to rounding."""
    from snag_amd import ops
    from snag_amd._lib import call, ptr, stream
    N, R, D, B, K = 300, 5, 128, 512, 32
    ent, rel, h, r, t, _ = _mk(N, R, D, B, K, seed=3, idt=torch.int32)
    g = torch.Generator().manual_seed(4)
    neg = torch.randint(0, 8, (B, K), generator=g, dtype=torch.int32)
    d = [x.to(DEV) for x in (ent, rel, h, r, t, neg)]
    de_ref, dr_ref = ops.score_margin_bwd(*d, 1.5)

    def fixed():
        acc_e = torch.zeros(N, D, device=DEV, dtype=torch.int64)
        acc_r = torch.zeros(R, D, device=DEV, dtype=torch.int64)
        call("snag_score_margin_bwd_fixed", ptr(d[0]), D, N, ptr(d[1]), R, D, ptr(d[2]), ptr(d[3]), ptr(d[4]), ptr(d[5]), 4, B, K, 1.5, 1.0, None, None,
             ptr(acc_e), ptr(acc_r), stream())
        de = torch.full((N, D + 8), 7.0, device=DEV)[:, :D]
        dr = torch.empty(R, D, device=DEV)
        call("snag_fixed_to_float", ptr(acc_e), N, D, ptr(de), de.stride(0), 0, stream())
        call("snag_fixed_to_float", ptr(acc_r), R, D, ptr(dr), D, 0, stream())
        return de.clone(), dr
    a, b = fixed(), fixed()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert_close(a[0], de_ref, atol=1e-5 * float(de_ref.abs().max()), rtol=1e-5)
    assert_close(a[1], dr_ref, atol=1e-5 * float(dr_ref.abs().max()), rtol=1e-5)


@pytest.mark.parametrize("N,R,D,B,K,masked", [(700, 9, 128, 100, 32, True), (700, 9, 128, 100, 32, False), (333, 3, 64, 40, 5, True), (1000, 5, 256, 64, 16, True),
                                              (17, 1, 8, 5, 2, True), (15000, 558, 128, 1024, 32, True)])
def test_score_segmented_entry_with_layernorm_backward(N, R, D, B, K, masked):
    """snag_score_margin_fwd_bwd_seg_ln: the scorer's table is LayerNorm(t) and the per-destination pass carries the gradient through the
    LayerNorm itself.  Scores / loss / d_rel equal the plain segmented entry's bits; dt, dtm and the three column sums equal torch autograd
    through layer_norm (fp64) on the same table (model/SNAG_tools.py:218-222 backward; model/MSNEA.py:110-122)."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(N + D + K)
    t_pre = torch.randn(N, D, generator=g)
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    rel = torch.randn(R, D, generator=g) * 0.5
    h, r, t = (torch.randint(0, hi, (B,), generator=g, dtype=torch.int32) for hi in (N, R, N))
    neg = torch.randint(0, N, (B, K), generator=g, dtype=torch.int32)
    keep = (torch.rand(N, D, generator=g) > 0.1).to(torch.uint8) if masked else None
    dev = [x.to(DEV) for x in (t_pre, gamma, beta, rel, h, r, t, neg)]
    loss, pos, ng, dt, dtm, dgam, dbet, dcol, d_rel, ent = ops.score_margin_fwd_bwd_seg_ln(*dev, 6.0, keep=keep.to(DEV) if masked else None, keep_p=0.1)
    # the plain entry on the same table: same scores / loss / relation gradient, bit for bit
    l2, p2, n2, _, d_ent, d_rel2 = ops.score_margin_fwd_bwd_seg(ent, dev[3], *dev[4:], 6.0)
    assert torch.equal(pos, p2) and torch.equal(ng, n2) and torch.equal(loss, l2) and torch.equal(d_rel, d_rel2)
    # LayerNorm backward of the plain entry's entity gradient, fp64 autograd
    tp = t_pre.double().requires_grad_(True)
    gm, bt = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(tp, (D,), gm, bt, 1e-12)
    y.backward(d_ent.cpu().double())
    scale = max(1.0, float(tp.grad.abs().max()))
    assert_close(dt, tp.grad, atol=2e-5 * scale, rtol=1e-5, what="dt")
    assert_close(dgam, gm.grad, atol=2e-4 * max(1.0, float(gm.grad.abs().max())), rtol=1e-4, what="dgamma")
    assert_close(dbet, bt.grad, atol=2e-4 * max(1.0, float(bt.grad.abs().max())), rtol=1e-4, what="dbeta")
    ref_m = tp.grad * keep.double() / 0.9 if masked else tp.grad
    if masked:
        assert_close(dtm, ref_m, atol=2e-5 * scale / 0.9, rtol=1e-5, what="dtm")
    assert_close(dcol, ref_m.sum(0), atol=2e-4 * max(1.0, float(ref_m.sum(0).abs().max())), rtol=1e-4, what="column sums")
